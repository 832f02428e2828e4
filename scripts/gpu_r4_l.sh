#!/bin/bash
# A/B: cfg4 with the pooled spectrogram from k_features (conv launch without its fused STFT phase) vs from the fused conv kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
timeout 600 python -m pytest tests -m gpu -q -x -k "feature or cfg4 or context" > "$OUT/pytest_l.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_l.log"
cp sound-spaces_amd/csrc/libss_hip.so /tmp/libss_hip.product.so
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_AB ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error")
trap 'cp /tmp/libss_hip.product.so "$GRAFT_REPO_ROOT/sound-spaces_amd/csrc/libss_hip.so"' EXIT
for ROUND in 1 2; do
  for V in late fused; do
    if [ $V = fused ]; then export SS_HIP_FEAT_KEEP_FUSED=1; else unset SS_HIP_FEAT_KEEP_FUSED; fi
    timeout 600 python bench.py --config cfg4 --no-cpu-baseline --no-plugin-path > "$OUT/bench_cfg4_$V.json" 2>/dev/null
    python - $V "$OUT" <<'PY'
import json,sys
d=json.loads(open(f'{sys.argv[2]}/bench_cfg4_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'value', round(d['value']/1e6,3), 'ms_per_step', d['ms_per_step'], 'kernel avg', d['roofline'].get('avg_launch_ms'))
PY
  done
done

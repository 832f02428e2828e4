#!/bin/bash
# A/B: window pairs of the fused kernels' hand-off in registers (product) vs read from LDS per block (-DSS_NO_WINREG)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
cp sound-spaces_amd/csrc/libss_hip.so /tmp/libss_hip.product.so
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_NO_WINREG ss_hip.hip -o /tmp/libss_hip.old.so 2>&1 | grep -E "error")
trap 'cp /tmp/libss_hip.product.so "$GRAFT_REPO_ROOT/sound-spaces_amd/csrc/libss_hip.so"' EXIT
for ROUND in 1 2 3; do
  for V in product old; do
    cp /tmp/libss_hip.$V.so sound-spaces_amd/csrc/libss_hip.so
    timeout 600 python bench.py --no-cpu-baseline --no-plugin-path > "$OUT/bench_winreg_$V.json" 2>/dev/null
    python - $V "$OUT" <<'PY'
import json,sys
d=json.loads(open(f'{sys.argv[2]}/bench_winreg_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'value', round(d['value']/1e6,3), 'ms_per_step', d['ms_per_step'], 'kernel avg us', round(1e3*d['roofline']['avg_launch_ms'],2), 'single', round(d['ctx_single_stream']['value']/1e6,3), 'spectral', round(d['spectral_bank']['value']/1e6,3))
PY
  done
done

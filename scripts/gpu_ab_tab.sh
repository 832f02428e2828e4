#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/tab
# the SS_HIP_* A/B switches exist only in -DSS_AB builds (the product library reads no environment variables): build one
# on the box for this script; the in-tree product library is restored at the end
cp sound-spaces_amd/csrc/libss_hip.so /tmp/libss_hip.product.so
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_AB ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error")
trap 'cp /tmp/libss_hip.product.so "$GRAFT_REPO_ROOT/sound-spaces_amd/csrc/libss_hip.so"' EXIT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/tab/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/tab/pytest.log
for V in tab notab tab notab; do
  if [ $V = notab ]; then export SS_HIP_NO_UNIT_TAB=1; else unset SS_HIP_NO_UNIT_TAB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-plugin-path > gpurun_out/tab/bench_$V.json 2>/dev/null
  python - $V <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/tab/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'value',round(d['value']/1e6,3),'ms',d['ms_per_step'], 'ctx_single',round(d['ctx_single_stream']['value']/1e6,3), d['ctx_single_stream']['ms_per_step'], 'preplanned', round(d['preplanned_single_stream']['value']/1e6,3), 'spectral', round(d['spectral_bank']['value']/1e6,3))
PY
done

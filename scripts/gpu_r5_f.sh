#!/bin/bash
# pass F: tests; sorted unit table A/B again (32-bit keys; value, sustained rate and host time per call); loader bench with the
# adaptive prefetch and the walks at 8 / 128 envs; SS2.0 deferred: live columns vs the request walk on the same box.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r5f"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
cp sound-spaces_amd/csrc/libss_hip.so /tmp/libss_hip.product.so
cp gpurun_in/libss_hip_ab.so sound-spaces_amd/csrc/libss_hip.so
for rep in 1 2 3; do
  for V in sort nosort; do
    if [ $V = nosort ]; then export SS_HIP_NO_SORT=1; else unset SS_HIP_NO_SORT; fi
    timeout 300 python bench.py --no-cpu-baseline --no-plugin-path --no-secondary > "$OUT/ab_${V}_$rep.json" 2>/dev/null
    python - "$OUT/ab_${V}_$rep.json" $V <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2],'value',round(d['value']/1e6,3),'ms',d['ms_per_step'],'sustained',round(d['sustained']['value']/1e6,3),'host_us',d.get('host_us_per_call'),'preplanned',d['preplanned_single_stream']['ms_per_step'])
PY
  done
done
unset SS_HIP_NO_SORT
cp /tmp/libss_hip.product.so sound-spaces_amd/csrc/libss_hip.so
timeout 900 python scripts/bench_loader.py --out "$OUT/loader.json" > "$OUT/loader.log" 2>&1; echo "loader rc=$?"; grep -v amdgpu "$OUT/loader.log" | grep -v '"files"' | cut -c1-400
for i in 1 2; do
  timeout 300 python scripts/bench_deferred_continuous.py 2>/dev/null | cut -c1-330
  timeout 300 python scripts/bench_deferred_continuous.py --walk 2>/dev/null | cut -c1-330
done
for C in cfg1 cfg3; do
  timeout 300 python bench.py --config $C --no-plugin-path --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C','value',d['value'],'ms',d['ms_per_step'],'host_us',d.get('host_us_per_call'),'kernel',d['roofline']['avg_launch_ms'])"
done

#!/bin/bash
# Round-4 final pass: GPU test-suite, the profile pass (scripts/gpu_profile_r4.sh), then - with the pass's traffic.json in place
# under profiles/r4/ - the bench lines again so that roofline.traffic is filled in, the boundary modes, the eager profile, the
# feature kernels and the FFT-core A/B.  Everything lands in gpurun_out/prof_r4/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r4"
timeout 1500 python -m pytest tests -m gpu -q > /tmp/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -4 /tmp/pytest_final.log
bash scripts/gpu_profile_r4.sh > /tmp/profile_pass.log 2>&1; echo "profile rc=$?"
cp /tmp/pytest_final.log "$OUT/pytest_gpu.log"; cp /tmp/profile_pass.log "$OUT/profile_pass.log"
mkdir -p profiles/r4; cp "$OUT/traffic.json" profiles/r4/traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_headline_driver_protocol.json" 2> "$OUT/bench_headline_driver_protocol.err"; echo "driver-protocol rc=$?"
timeout 900 python bench.py > "$OUT/bench_headline.json" 2> "$OUT/bench_headline.err"; echo "headline rc=$?"
timeout 900 python bench.py --config cfg2 --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"
timeout 900 python bench.py --config cfg4 --steps 100 --no-cpu-baseline > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"
timeout 600 python scripts/bench_boundary.py > "$OUT/bench_boundary.jsonl" 2> "$OUT/bench_boundary.err"; echo "boundary rc=$?"
timeout 300 python scripts/prof_eager.py > "$OUT/prof_eager.txt" 2>&1; echo "eager rc=$?"; grep "^eager" "$OUT/prof_eager.txt"
timeout 300 python scripts/kbench_features.py > "$OUT/kbench_features.json" 2>/dev/null; cat "$OUT/kbench_features.json"
timeout 600 python scripts/kbench32.py --out "$OUT/kbench32.json" > "$OUT/kbench32.log" 2>&1; tail -3 "$OUT/kbench32.log"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prof_r4/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value',d['value'], 'ms',d['ms_per_step'], 'roofline',d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'traffic', d['roofline'].get('traffic'))
    except Exception as e:
        print(f,'ERR',e)
PY

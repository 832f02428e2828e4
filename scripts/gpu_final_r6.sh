#!/bin/bash
# Round-6 final pass (ONE script; run through gpurun): GPU test-suite, the profile pass (scripts/gpu_profile_r6.sh: one bench line
# WITH its cpu_baseline + one rocprofv3 kernel trace per configuration, PMC for headline / cfg1 / cfg2 / cfg4 / 10 envs @44.1 kHz,
# counter calibration), then - with that pass's traffic.json in place under profiles/r6/ - the headline lines again so that
# roofline.traffic is filled in; the savi pre-training set (scripts/bench_dataset.py), the RIR miss path (scripts/bench_loader.py),
# SS2.0 deferred mode, the boundary modes, the eager profile, the feature kernels.  Everything lands in gpurun_out/prof_r6/ (copy to
# profiles/r6/).  The conv-kernel roofline table comes from scripts/gpu_conv_roofline_r6.sh, the same-box A/B files of the round
# (kbench_blocks_44k.txt) from scripts/gpu_obs_blocks_r6.sh - both build / use what they need themselves.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r6"
timeout 1500 python -m pytest tests -m gpu -q > /tmp/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -4 /tmp/pytest_final.log
timeout 300 python -m pytest tests/test_context.py -m gpu -q -s -k "observe_features_equals" 2>&1 | grep "vs oracle" > /tmp/features_vs_oracle.txt; cat /tmp/features_vs_oracle.txt
bash scripts/gpu_profile_r6.sh > /tmp/profile_pass.log 2>&1; echo "profile rc=$?"
cp /tmp/pytest_final.log "$OUT/pytest_gpu.log"; cp /tmp/profile_pass.log "$OUT/profile_pass.log"; cp /tmp/features_vs_oracle.txt "$OUT/features_vs_oracle.txt"
mkdir -p profiles/r6; cp "$OUT/traffic.json" profiles/r6/traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_headline_driver_protocol.json" 2> "$OUT/bench_headline_driver_protocol.err"; echo "driver-protocol rc=$?"
timeout 900 python bench.py > "$OUT/bench_headline.json" 2> "$OUT/bench_headline.err"; echo "headline rc=$?"
timeout 900 python bench.py --rir-bank time --no-plugin-path > "$OUT/bench_headline_time.json" 2> "$OUT/bench_headline_time.err"
timeout 900 python bench.py --config cfg2 --steps 40 --warmup 5 --no-plugin-path > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"
timeout 900 python bench.py --config cfg4 --steps 100 --no-plugin-path > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"
timeout 900 python scripts/bench_dataset.py > "$OUT/dataset.json" 2> "$OUT/dataset.err"; echo "dataset rc=$?"; cat "$OUT/dataset.json"
timeout 600 python scripts/bench_loader.py --out "$OUT/loader.json" > "$OUT/loader.log" 2>&1; echo "loader rc=$?"
# SS2.0 deferred: the live-column path and, on the SAME box, round 4's per-request walk (alternating: host speed differs box to box)
: > "$OUT/bench_deferred_continuous.jsonl"
for i in 1 2 3; do
  timeout 300 python scripts/bench_deferred_continuous.py >> "$OUT/bench_deferred_continuous.jsonl" 2> /dev/null
  timeout 300 python scripts/bench_deferred_continuous.py --walk >> "$OUT/bench_deferred_continuous.jsonl" 2> /dev/null
  timeout 300 python scripts/bench_deferred_continuous.py --scatter-copy >> "$OUT/bench_deferred_continuous.jsonl" 2> /dev/null
done
cut -c1-330 "$OUT/bench_deferred_continuous.jsonl"
timeout 600 python scripts/bench_boundary.py > "$OUT/bench_boundary.jsonl" 2> "$OUT/bench_boundary.err"; echo "boundary rc=$?"
timeout 300 python scripts/prof_eager.py > "$OUT/prof_eager.txt" 2>&1; echo "eager rc=$?"; grep "^eager" "$OUT/prof_eager.txt"
timeout 300 python scripts/kbench_features.py > "$OUT/kbench_features.json" 2>/dev/null; cat "$OUT/kbench_features.json"
for r in 0.01 0.05 0.25; do timeout 200 python scripts/miss_breakdown.py --rate $r 2>&1 | grep -v amdgpu | tail -12 >> "$OUT/miss_breakdown.txt"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prof_r6/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value',d['value'], 'ms',d['ms_per_step'], 'roofline',d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'traffic', (d['roofline'].get('traffic') or {}).get('bytes') if isinstance(d['roofline'].get('traffic'),dict) else d['roofline'].get('traffic'), 'cpu', d.get('cpu_baseline',{}).get('value'), 'x', d.get('speedup_vs_cpu_all_cores'))
    except Exception as e:
        print(f,'ERR',e)
PY
grep -h miss_rate "$OUT/loader.log" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['mode'],d['reader'],d['miss_rate'],d['trainer_half_us_per_step_median'],d['env_steps_per_s_trainer_half'])"
grep -h '\"files\"' "$OUT/loader.log"

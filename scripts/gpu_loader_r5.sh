#!/bin/bash
# The RIR miss path alone (after a host-side change that leaves the kernels - and traffic.json - untouched): loader.json /
# loader.log / miss_breakdown.txt of gpurun_out/prof_r5
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r5"; mkdir -p "$OUT"
timeout 120 python -m pytest tests/test_wav_loader.py tests/test_deferred_columns.py tests/test_deferred.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python scripts/bench_loader.py --out "$OUT/loader.json" > "$OUT/loader.log" 2>&1; echo "loader rc=$?"
: > "$OUT/miss_breakdown.txt"
for r in 0.01 0.05 0.25; do timeout 200 python scripts/miss_breakdown.py --rate $r 2>/dev/null | grep -v "^\[" >> "$OUT/miss_breakdown.txt"; done
timeout 200 python scripts/miss_breakdown.py --rate 0.05 --full-store 2>/dev/null | grep -v "^\[" >> "$OUT/miss_breakdown.txt"
cat "$OUT/miss_breakdown.txt"
grep -h miss_rate "$OUT/loader.log" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['mode'],d['reader'],d['miss_rate'],d.get('store'),d['trainer_half_us_per_step_median'],d['env_steps_per_s_trainer_half'])"
grep -h '\"files\"' "$OUT/loader.log"; grep -h walk "$OUT/loader.log" | cut -c1-400

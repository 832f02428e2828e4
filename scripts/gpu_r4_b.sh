#!/bin/bash
# round 4, second GPU pass: GPU test-suite (deferred column path, core32), boundary modes, the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q -x > "$OUT/pytest_b.log" 2>&1; echo "pytest rc=$?"; tail -8 "$OUT/pytest_b.log"
timeout 600 python scripts/bench_boundary.py > "$OUT/bench_boundary_b.jsonl" 2> "$OUT/bench_boundary_b.err"; echo "boundary rc=$?"; cat "$OUT/bench_boundary_b.jsonl"; tail -3 "$OUT/bench_boundary_b.err"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_b.json" 2> "$OUT/bench_b.err"; echo "bench rc=$?"; tail -3 "$OUT/bench_b.err"
python - <<'PY'
import json,os
j=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r4/bench_b.json")).read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "roofline", j["roofline"]["frac"])
print(json.dumps(j["plugin_path"]["deferred"], indent=1))
PY

#!/bin/bash
# round 4, fourth GPU pass: the one-pass feature kernel - parity, kernel timings, cfg4 with the features in the timed region
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_features.py tests/test_logmel.py tests/test_gccphat.py tests/test_context.py -m gpu -q -x > "$OUT/pytest_d.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_d.log"
timeout 300 python scripts/kbench_features.py > "$OUT/kbench_features_d.json" 2> "$OUT/kbench_features_d.err"; echo "kbench rc=$?"; cat "$OUT/kbench_features_d.json"; tail -2 "$OUT/kbench_features_d.err"
timeout 900 python bench.py --config cfg4 --no-cpu-baseline > "$OUT/bench_cfg4_d.json" 2> "$OUT/bench_cfg4_d.err"; echo "cfg4 rc=$?"; tail -3 "$OUT/bench_cfg4_d.err"
timeout 900 python bench.py --config cfg4 --features none --no-cpu-baseline > "$OUT/bench_cfg4_nofeat_d.json" 2>> "$OUT/bench_cfg4_d.err"; echo "cfg4 nofeat rc=$?"
python - <<'PY'
import json,os
for f in ("bench_cfg4_d.json","bench_cfg4_nofeat_d.json"):
    j=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r4",f)).read().strip().splitlines()[-1])
    print(f, "value", j["value"], "ms/step", j["ms_per_step"], "roofline", j["roofline"]["frac"], "avg_launch_ms", j["roofline"]["avg_launch_ms"], "single", j["preplanned_single_stream"]["ms_per_step"], j.get("ctx_single_stream",{}).get("ms_per_step"))
PY

#!/bin/bash
# round 4, third GPU pass: GPU test-suite after the ADVICE / unit-table refactor, the driver's bench line with 25 regions
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q -x > "$OUT/pytest_c.log" 2>&1; echo "pytest rc=$?"; tail -8 "$OUT/pytest_c.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_c_driver.json" 2> "$OUT/bench_c.err"; echo "bench rc=$?"; tail -3 "$OUT/bench_c.err"
timeout 900 python bench.py --no-cpu-baseline > "$OUT/bench_c_default.json" 2>> "$OUT/bench_c.err"; echo "bench default rc=$?"
python - <<'PY'
import json,os
for f in ("bench_c_driver.json","bench_c_default.json"):
    j=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r4",f)).read().strip().splitlines()[-1])
    print(f, "value", j["value"], "ms/step", j["ms_per_step"], "roofline", j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], "spread", j.get("value_spread"))
    print("   preplanned", j["preplanned_single_stream"]["value"], "conv_only", j.get("roofline_conv_only",{}).get("frac"))
    d=j["plugin_path"]["deferred"]
    print("   deferred", {k:(v if not isinstance(v,dict) else {a:b for a,b in v.items()}) for k,v in d.items() if k!="note"})
    print("   bound_sims", j["plugin_path"]["bound_sims"]["env_steps_per_s"], "columns", j["plugin_path"]["columns"]["env_steps_per_s"])
PY

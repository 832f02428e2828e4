cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
t0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b20.json 2> gpurun_out/b20.err
t1=$(date +%s.%N)
echo "bench elapsed $(echo "$t1 - $t0" | bc) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/b20.json").read().strip().splitlines()[-1])
print(d["metric"]); print(d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["dtype"], d["vs_baseline"])
print(d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"]); print(d["config"]["workload"][:300])
PY
t0=$(date +%s.%N)
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
t1=$(date +%s.%N)
echo "build+smoke elapsed $(echo "$t1 - $t0" | bc) s"

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r6b"; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_dataset.py tests/test_context.py tests/test_c_abi.py -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_gpu.log"
timeout 900 python scripts/bench_dataset.py > "$OUT/dataset.json" 2> "$OUT/dataset.err"; echo "dataset rc=$?"; cat "$OUT/dataset.json"; tail -3 "$OUT/dataset.err"
timeout 900 python bench.py --no-plugin-path --no-cpu-baseline > "$OUT/bench_headline.json" 2> "$OUT/bench_headline.err"; echo "bench rc=$?"
timeout 600 python bench.py --config cfg1 --no-plugin-path --no-cpu-baseline > "$OUT/bench_cfg1.json" 2> "$OUT/bench_cfg1.err"; echo "cfg1 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6b/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value',d['value'], 'ms',d['ms_per_step'], 'host_us', d.get('host_us_per_call'), 'pipelined', d['pipelined']['value'], 'dep', {k:v.get('value') for k,v in d['dependent'].items() if isinstance(v,dict)})
    except Exception as e:
        print(f,'ERR',e)
PY

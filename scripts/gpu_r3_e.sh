#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r3e"; mkdir -p "$OUT"
timeout 600 python bench.py --no-cpu-baseline --no-plugin-path > "$OUT/bench_headline_default.json" 2> "$OUT/bench_headline_default.err"; echo "b2 rc=$?"; tail -3 "$OUT/bench_headline_default.err"
timeout 600 python bench.py --no-cpu-baseline --no-plugin-path --steps 20 --warmup 5 > "$OUT/bench_headline_s20.json" 2> "$OUT/bench_headline_s20.err"; echo "b1 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'value',d['value'], 'ms',d['ms_per_step'], 'roofline',d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'pipe',d['roofline']['pipeline_frac'])
        for k,v in d.items():
            if isinstance(v,dict) and 'value' in v and k not in('roofline','cpu_baseline'): print('   ',k,v.get('value'),v.get('ms_per_step'),v.get('avg_launch_ms'))
    except Exception as e:
        print(f,'ERR',e)
PY

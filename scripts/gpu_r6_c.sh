#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r6c"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -8 "$OUT/pytest_gpu.log"
timeout 600 python bench.py --config cfg1 --no-plugin-path --no-cpu-baseline > "$OUT/bench_cfg1.json" 2> "$OUT/bench_cfg1.err"; echo "cfg1 rc=$?"
timeout 600 python bench.py --config cfg3 --no-plugin-path --no-cpu-baseline > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"; echo "cfg3 rc=$?"
timeout 600 python scripts/bench_boundary.py > "$OUT/bench_boundary.jsonl" 2> "$OUT/bench_boundary.err"; echo "boundary rc=$?"; cat "$OUT/bench_boundary.jsonl" | cut -c1-600
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c/bench_cfg*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['config']['rir_bank'], 'value',d['value'], 'ms',d['ms_per_step'], 'kernel_ms', d['roofline']['avg_launch_ms'], 'host_us', d.get('host_us_per_call'), 'pipelined', d['pipelined']['value'], 'dep', {k:v.get('value') for k,v in d['dependent'].items() if isinstance(v,dict)}, 'other', {k:d[k].get('value') for k in ('spectral_bank','time_domain_bank') if k in d})
    except Exception as e:
        print(f,'ERR',e)
PY

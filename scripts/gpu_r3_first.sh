#!/bin/bash
# round 3, first GPU pass: parity of the fused 44.1 kHz kernel, then first timings (headline untouched, 44.1 kHz 128 units, cfg2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
timeout 300 python bench.py --sr 44100 --envs 128 --no-cpu-baseline --no-plugin-path --steps 50 --warmup 5 > gpurun_out/r3a/bench_44k_128.json 2> gpurun_out/r3a/bench_44k_128.err; echo "b1 rc=$?"
timeout 300 python bench.py --config cfg2 --no-cpu-baseline --no-plugin-path --steps 30 --warmup 5 > gpurun_out/r3a/bench_cfg2.json 2> gpurun_out/r3a/bench_cfg2.err; echo "b2 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3a/bench_headline.json 2> gpurun_out/r3a/bench_headline.err; echo "b3 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], {k:(v.get('value'),v.get('ms_per_step')) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k not in('roofline','cpu_baseline')}, d.get('roofline_conv_only',{}).get('avg_launch_ms'))
    except Exception as e:
        print(f, 'ERR', e)
PY

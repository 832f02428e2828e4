#!/bin/bash
# round 6, first pass on the box: GPU tests, the new bench line (dependent-step headline) and the conv-kernel roofline table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r6a"; mkdir -p "$OUT"
timeout 1200 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
timeout 900 python bench.py > "$OUT/bench_headline.json" 2> "$OUT/bench_headline.err"; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-plugin-path > "$OUT/bench_headline_driver_protocol.json" 2> "$OUT/bench_driver.err"; echo "bench20 rc=$?"
timeout 600 python bench.py --config cfg1 --no-plugin-path --no-cpu-baseline > "$OUT/bench_cfg1.json" 2> "$OUT/bench_cfg1.err"; echo "cfg1 rc=$?"
timeout 600 python bench.py --config cfg2 --steps 40 --warmup 5 --no-plugin-path --no-cpu-baseline > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"; echo "cfg2 rc=$?"
timeout 600 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-plugin-path --no-cpu-baseline > "$OUT/bench_gpus2_gloo.json" 2> "$OUT/bench_gpus2.err"; echo "gpus2 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value',d['value'], 'ms',d['ms_per_step'], 'roofline',d['roofline']['frac'], d['roofline']['avg_launch_ms'], {k:(v.get('value') if 'value' in v else {kk:vv.get('value') for kk,vv in v.items() if isinstance(vv,dict)}) for k,v in d.items() if isinstance(v,dict) and k not in ('roofline','cpu_baseline','config','value_spread','sustained','gpu_ms_per_step','roofline_valu','plugin_path')})
        for k in d:
            if k.startswith('roofline_conv'): print('   ',k,d[k])
    except Exception as e:
        print(f,'ERR',e)
PY
bash scripts/gpu_conv_roofline_r6.sh > "$OUT/conv_roofline.log" 2>&1
cp gpurun_out/conv_roofline/conv_roofline.txt gpurun_out/conv_roofline/conv_roofline.json "$OUT"/ 2>/dev/null
cat "$OUT/conv_roofline.txt"

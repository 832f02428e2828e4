#!/bin/bash
# the tracked bench lines only (no rocprof): every preset as the driver would run it + the driver's own protocol, with
# profiles/r3/traffic.json of this build in place (so that roofline.traffic is filled in)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/lines; rm -rf $OUT; mkdir -p $OUT
declare -A CFG
CFG[headline]=""; CFG[cfg1]="--config cfg1"; CFG[cfg2]="--config cfg2 --steps 40 --warmup 5"; CFG[cfg4]="--config cfg4 --steps 100"
CFG[replica44k_128]="--sr 44100 --envs 128 --steps 60 --warmup 5"
for NAME in headline cfg1 cfg2 cfg4 replica44k_128; do
  EXTRA="--no-cpu-baseline"; [ "$NAME" = headline ] && EXTRA=""
  timeout 900 python bench.py ${CFG[$NAME]} $EXTRA > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.err || echo "bench $NAME failed"
done
for i in 1 2 3; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_s20_$i.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/lines/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], round(d['value']/1e6,3), round(1e3*d['ms_per_step'],2), r['frac'], round(1e3*r['avg_launch_ms'],2), r.get('traffic'), r.get('pipeline_frac'),
          {k: round(v['value']/1e6,3) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k not in ('roofline','cpu_baseline')})
PY

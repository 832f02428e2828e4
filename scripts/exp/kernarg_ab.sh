cd "$GRAFT_REPO_ROOT"
for R in 1 2; do
for V in unset 0 1; do
  if [ $V = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$V; fi
  timeout 600 python bench.py --no-cpu-baseline --no-plugin-path > /tmp/b_$V.json 2>/dev/null
  python - $V <<'PY'
import json,sys
d=json.loads(open(f'/tmp/b_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print('HIP_FORCE_DEV_KERNARG', sys.argv[1], 'value',round(d['value']/1e6,3),'ms',d['ms_per_step'], 'ctx_single',round(d['ctx_single_stream']['value']/1e6,3), 'preplanned', round(d['preplanned_single_stream']['value']/1e6,3), 'kernel us', round(1e3*d['roofline']['avg_launch_ms'],2))
PY
done
done

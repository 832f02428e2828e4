#!/bin/bash
# The length-bucket mix (scripts/bench_buckets.py) and the multi-rank flows of bench.py exercised on the ONE GPU of a gpurun
# box: bench.py --gpus N launches its N ranks itself; both / all ranks share the GPU and gloo is the control plane, so this is
# FUNCTIONAL only (rank flow, every exchange transport incl. the consumer-release protocol, strong scaling of cfg[3]).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT="$GRAFT_REPO_ROOT/gpurun_out/dist_smoke"; mkdir -p "$OUT"
timeout 600 python scripts/bench_buckets.py > "$OUT/bench_buckets.json" 2> "$OUT/bench_buckets.err"; echo "buckets rc=$?"; cat "$OUT/bench_buckets.json"; tail -3 "$OUT/bench_buckets.err"
# multi-rank flows of bench.py on the ONE GPU of the box (functional only: both ranks share the GPU; gloo control plane)
for EX in allgather peercopy gather none; do
  timeout 600 python bench.py --gpus 2 --backend gloo --exchange $EX --steps 24 --warmup 4 --spinup-steps 64 --envs 32 --bank-mib 64 --no-secondary > "$OUT/dist2_$EX.json" 2> "$OUT/dist2_$EX.err"; echo "dist $EX rc=$?"
  python -c "
import json,sys
try:
    d=json.loads(open('$OUT/dist2_$EX.json').read().strip().splitlines()[-1]); print('$EX', d['n_gpus'], d['value'], d['config']['exchange'])
except Exception as e: print('$EX ERR', e); print(open('$OUT/dist2_$EX.err').read()[-1500:])
"
done
timeout 600 python bench.py --gpus 8 --backend gloo --scaling strong --envs 128 --steps 16 --warmup 4 --spinup-steps 32 --bank-mib 64 --no-secondary > "$OUT/dist8_strong.json" 2> "$OUT/dist8_strong.err"; echo "dist8 rc=$?"
python -c "
import json
try:
    d=json.loads(open('$OUT/dist8_strong.json').read().strip().splitlines()[-1]); print('strong8', d['n_gpus'], d['value'], d['config']['envs_per_gpu'], d['config']['exchange'])
except Exception as e: print('strong8 ERR', e); print(open('$OUT/dist8_strong.err').read()[-1500:])
"

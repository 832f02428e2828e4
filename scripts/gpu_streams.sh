#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for S in 1 2 3; do
  echo "== 1 GPU, --streams $S"; timeout 300 python bench.py --no-cpu-baseline --streams $S 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['config']['streams'])"
done
echo "== 2 ranks gloo shared GPU (auto streams)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 64 --warmup 5 --backend gloo --bank-mib 128 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['streams'], j['config']['exchange'])"

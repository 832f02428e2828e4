#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
echo "== 2 ranks, gloo, shared GPU, allgather"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 --backend gloo --bank-mib 128 2>&1 | tail -3
echo "== 2 ranks, gloo, exchange none"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 50 --warmup 5 --backend gloo --bank-mib 128 --exchange none 2>&1 | tail -2
echo "== 1 rank nccl init path (world 1 via torchrun)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --bank-mib 128 2>&1 | tail -1 | cut -c1-300
echo "== cfg[3] shape, functionally: 8 ranks x 16 envs (strong scaling of 128 envs), gloo, all ranks on the one GPU, chunked gather"
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 8 --scaling strong --steps 30 --warmup 5 --backend gloo --bank-mib 64 2>&1 | tail -2 | cut -c1-900

#!/bin/bash
# Runs on the GPU box via gpurun: smoke + GPU parity tests + bench + rocprofv3 kernel trace.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5
echo "== pytest gpu"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
echo "== bench default"; timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench_default.json
for E in 32 512 2048; do
  echo "== bench envs=$E"; timeout 300 python bench.py --envs $E --no-cpu-baseline --steps 100 2>&1 | tail -1 | tee gpurun_out/bench_envs$E.json
done
echo "== bench 44.1k"; timeout 300 python bench.py --sr 44100 --envs 128 --no-cpu-baseline --steps 50 --bank-mib 768 2>&1 | tail -1 | tee gpurun_out/bench_44k.json
echo "== rocprof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r1" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps 200 2>&1 | tail -3 )
find gpurun_out/prof_r1 -name "*stats*" | head; 
for f in $(find gpurun_out/prof_r1 -name "*kernel_stats*.csv" | head -2); do echo "-- $f"; head -12 "$f"; done

#!/bin/bash
# Round-end style GPU pass: smoke, parity tests, bench (with CPU baseline), kernel trace + PMC of the bench command.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r1}
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== pytest gpu"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_$TAG.json
echo "== bench 44.1k"; timeout 600 python bench.py --sr 44100 --no-cpu-baseline --steps 50 --bank-mib 768 2>&1 | tail -1 | tee gpurun_out/bench_${TAG}_44k.json
echo "== bench 32 envs (configs[1])"; timeout 600 python bench.py --envs 32 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_${TAG}_32env.json
echo "== kbench"; timeout 300 python scripts/kbench.py --sizes 32,128,512,2048 2>&1 | grep "^N=" | tee gpurun_out/kbench_$TAG.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 200"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG" -o trace -- $CMD > /dev/null 2>&1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG" -o pmc$i -- $CMD > /dev/null 2>&1 || echo "pmc pass $i failed"
done
cd "$GRAFT_REPO_ROOT"
python scripts/prof_summary.py gpurun_out/prof_$TAG > /dev/null 2>&1
head -12 gpurun_out/prof_$TAG/trace_kernel_stats.csv

#!/bin/bash
# kernel trace + PMC passes for the bench command (counters in their own runs, no tracing domains mixed in)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r1}
echo "== pytest gpu"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5
echo "== kbench"; timeout 300 python scripts/kbench.py --sizes 128,2048 2>&1 | grep "^N="
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG" -o trace -- $CMD > /dev/null 2>&1
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_]+|LDSBankConflict|VALUBusy|MemUnitBusy)\b" | sort -u | tr '\n' ' ' | head -c 6000 > "$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/counters_avail.txt"
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG" -o pmc$i -- $CMD > /dev/null 2>&1 || echo "pmc pass $i failed"
done
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof_$TAG -type f | head -30
python scripts/prof_summary.py gpurun_out/prof_$TAG 2>&1 | tail -60

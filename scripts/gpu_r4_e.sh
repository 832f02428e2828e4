#!/bin/bash
# PMC passes on k_features<logmel,gccphat> (256 units): where does a round's time go
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4/pmc_feat"; rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/scripts/feat_only.py"
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT" -o pmc$i -- $CMD > /dev/null 2>&1 ) || echo "pmc pass $i failed"
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- $CMD > /dev/null 2>&1 ) || echo "trace failed"
python scripts/prof_summary.py "$OUT" | grep -v "^  .*k_conv\|torch" | head -60

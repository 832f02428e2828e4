#!/bin/bash
# PMC counters of the split-row kernel at cfg1 (32 envs: 4 parts per row) - where the 18-19 us of a split row go
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_small"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/bench.py --config cfg1 --no-cpu-baseline --no-plugin-path --no-secondary --streams 1 --spinup-steps 0 --regions 1 --sustain 0"
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_cfg1 -o pmc$i -- $CMD > /dev/null 2>&1 ) || echo "pmc pass $i failed"
done
python scripts/prof_summary.py /tmp/pmc_cfg1 > /dev/null 2>&1; cp /tmp/pmc_cfg1/summary.txt "$OUT/pmc_cfg1.txt"; cat "$OUT/pmc_cfg1.txt"

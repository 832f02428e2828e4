#!/bin/bash
# dynamic VALU instruction mix of the headline kernel (and cfg2): how much of SQ_INSTS_VALU is floating-point arithmetic
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/valu_mix"; rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z_0-9]*" | sort -u > "$OUT/avail.txt" )
cat "$OUT/avail.txt" | tr '\n' ' '; echo
for NAME in headline cfg2; do
  A=""; [ $NAME = cfg2 ] && A="--config cfg2"
  CMD="python $GRAFT_REPO_ROOT/bench.py $A --no-cpu-baseline --no-plugin-path --no-secondary --streams 1 --spinup-steps 0 --steps 40 --warmup 5"
  D="$OUT/pmc_$NAME"; i=0
  for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
             "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU SQ_INSTS_SMEM" \
             "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_GDS"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --pmc $PMC --output-format csv -d "$D" -o pmc$i -- $CMD > /dev/null 2>&1 ) || echo "pmc pass $i of $NAME failed"
  done
  python scripts/prof_summary.py "$D" > /dev/null 2>&1
  cp "$D/summary.txt" "$OUT/valu_mix_$NAME.txt"; rm -rf "$D"
  grep -A40 "k_conv<true\|k_obs_rows\|k_conv" "$OUT/valu_mix_$NAME.txt" | head -60
done

#!/bin/bash
# round 3, second GPU pass: where does k_obs_rows' time go?  ablation ladder (time-domain and spectral bank) + PMC pass
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r3b"; mkdir -p "$OUT"
bash scripts/gpu_rows_ladder.sh "--sr 44100 --sizes 128,512 --raw --only fused --reps 100 --bank-mib 1024" 1 2 4 6 8 16 31 > "$OUT/ladder_time.txt" 2>&1
bash scripts/gpu_rows_ladder.sh "--sr 44100 --sizes 128,512 --raw --only fused --reps 100 --bank-mib 1024 --spectral" 1 4 16 > "$OUT/ladder_spectral.txt" 2>&1
cat "$OUT/ladder_time.txt" "$OUT/ladder_spectral.txt"
for BANK in time spectral; do
  CMD="python $GRAFT_REPO_ROOT/bench.py --sr 44100 --envs 128 --no-cpu-baseline --no-plugin-path --no-secondary --steps 40 --warmup 5 --spinup-steps 0 --rir-bank $BANK"
  D="$OUT/pmc_$BANK"
  i=0
  for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d "$D" -o pmc$i -- $CMD > /dev/null 2>&1 ) || echo "pmc pass $i failed"
  done
  python scripts/prof_summary.py "$D" > /dev/null 2>&1
  cp "$D/summary.txt" "$OUT/pmc_summary_$BANK.txt"
  grep -A40 "PMC.*k_obs_rows" "$OUT/pmc_summary_$BANK.txt" | head -45
done

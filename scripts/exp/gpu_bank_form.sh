cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/bank_form; mkdir -p $OUT; : > $OUT/kbench_bank_form_16k.txt
for rep in 1 2; do
  for bank in "" "--spectral"; do
    echo "== rep=$rep bank=${bank:-time} fused, 16 kHz" >> $OUT/kbench_bank_form_16k.txt
    timeout 300 python scripts/kbench.py --sr 16000 --sizes 64,96,128,192,256,512,1024,2048 --only fused --raw --reps 100 --bank-mib 2048 $bank 2>/dev/null >> $OUT/kbench_bank_form_16k.txt
  done
done
cat $OUT/kbench_bank_form_16k.txt

#!/bin/bash
# Round 6: the same-box A/B runs behind profiles/r6/kbench_{bank_form_16k,l2warm_44k}.txt (kbench_blocks_44k.txt: gpu_obs_blocks_r6.sh).
# Variants are compile-time switches of the device code: the two libraries are built on the box and swapped in turn.
#   usage (through gpurun): bash scripts/gpu_ab_r6.sh bank_form | l2warm
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/ab_r6; mkdir -p $OUT
case "${1:-bank_form}" in
bank_form)   # fused 16-kHz kernel from the time-domain rows against the spectral rows, 64 ... 2048 units
  : > $OUT/kbench_bank_form_16k.txt
  for rep in 1 2; do
    for bank in "" "--spectral"; do
      echo "== rep=$rep bank=${bank:-time} fused, 16 kHz" >> $OUT/kbench_bank_form_16k.txt
      timeout 300 python scripts/kbench.py --sr 16000 --sizes 64,96,128,192,256,512,1024,2048 --only fused --raw --reps 100 --bank-mib 2048 $bank 2>/dev/null >> $OUT/kbench_bank_form_16k.txt
    done
  done
  cat $OUT/kbench_bank_form_16k.txt ;;
l2warm)      # k_obs_rows: the next block's RIR samples pulled into L2 under the STFT phase (-DSS_ROWS_NO_L2_WARM = off)
  cp sound-spaces_amd/csrc/libss_hip.so /tmp/lib_on.so
  (cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_ROWS_NO_L2_WARM ss_hip.hip -o /tmp/lib_off.so 2>&1 | grep -E "error")
  : > $OUT/kbench_l2warm_44k.txt
  for rep in 1 2 3; do
    for v in on off; do
      cp /tmp/lib_$v.so sound-spaces_amd/csrc/libss_hip.so
      for bank in "" "--spectral"; do
        echo "== rep=$rep l2_warm=$v bank=${bank:-time}" >> $OUT/kbench_l2warm_44k.txt
        timeout 300 python scripts/kbench.py --sr 44100 --sizes 64,128,512 --only fused --raw --reps 60 --bank-mib 2048 $bank 2>/dev/null >> $OUT/kbench_l2warm_44k.txt
      done
    done
  done
  cp /tmp/lib_on.so sound-spaces_amd/csrc/libss_hip.so
  cat $OUT/kbench_l2warm_44k.txt ;;
esac

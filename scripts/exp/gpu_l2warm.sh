cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/l2warm; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "44 or fused or rows or cfg2 or replica" 2>&1 | tail -3
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
cat $OUT/kbench_l2warm_44k.txt

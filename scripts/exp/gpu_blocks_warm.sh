cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/blocks_warm; mkdir -p $OUT
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ss_hip.hip -o /tmp/lib_on.so 2>&1 | grep -E "error")
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_BLOCKS_NO_L2_WARM ss_hip.hip -o /tmp/lib_off.so 2>&1 | grep -E "error")
: > $OUT/kbench_blocks_l2warm_44k.txt
for rep in 1 2 3; do
  for v in on off; do
    cp /tmp/lib_$v.so sound-spaces_amd/csrc/libss_hip.so
    for bank in "--spectral" ""; do
      echo "== rep=$rep l2_warm=$v bank=${bank:-time}" >> $OUT/kbench_blocks_l2warm_44k.txt
      timeout 300 python scripts/kbench.py --sr 44100 --sizes 1,5,10,32,42 --only fused --raw --reps 100 --bank-mib 1024 $bank 2>/dev/null >> $OUT/kbench_blocks_l2warm_44k.txt
    done
  done
done
cp /tmp/lib_on.so sound-spaces_amd/csrc/libss_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "44 or fused or rows or replica or split" 2>&1 | tail -2
cat $OUT/kbench_blocks_l2warm_44k.txt

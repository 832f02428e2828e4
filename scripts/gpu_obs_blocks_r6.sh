#!/bin/bash
# round 6: k_obs_blocks (one workgroup per output block of a row, small steps at 44.1 kHz) against k_obs_rows, same box, alternating.
# The product library reads no environment variables: a -DSS_AB build is made on the box (replaces the in-tree .so for this call only).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/obs_blocks"; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_AB ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error")
: > "$OUT/kbench_blocks_44k.txt"
for rep in 1 2; do
  for bank in "" "--spectral"; do
    for off in 0 1; do
      if [ $off = 1 ]; then export SS_HIP_NO_OBS_BLOCKS=1; else unset SS_HIP_NO_OBS_BLOCKS; fi
      echo "== rep=$rep bank=${bank:-time} obs_blocks=$((1-off))" >> "$OUT/kbench_blocks_44k.txt"
      timeout 300 python scripts/kbench.py --sr 44100 --sizes 1,5,10,16,32,42 --only fused --raw --reps 100 --bank-mib 1024 $bank 2>/dev/null >> "$OUT/kbench_blocks_44k.txt"
    done
  done
done
unset SS_HIP_NO_OBS_BLOCKS
cat "$OUT/kbench_blocks_44k.txt"

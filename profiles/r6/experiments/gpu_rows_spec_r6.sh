#!/bin/bash
# round 6: k_obs_rows_spec (a pair's 16 loads in flight, accumulator in registers, one row per workgroup) against
# k_obs_rows<SPECTRAL> (r3-r5), same box, alternating; -DSS_AB build made on the box (the product library reads no environment)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/rows_spec"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_AB ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error")
: > "$OUT/kbench_rows_spec_44k.txt"
for rep in 1 2; do
  for old in 0 1; do
    if [ $old = 1 ]; then export SS_HIP_OLD_SPEC_ROWS=1; else unset SS_HIP_OLD_SPEC_ROWS; fi
    echo "== rep=$rep spectral bank, k_obs_rows<SPECTRAL>(old)=$old" >> "$OUT/kbench_rows_spec_44k.txt"
    timeout 300 python scripts/kbench.py --sr 44100 --sizes 64,128,256,512 --only fused --raw --reps 60 --bank-mib 2048 --spectral 2>/dev/null >> "$OUT/kbench_rows_spec_44k.txt"
  done
done
unset SS_HIP_OLD_SPEC_ROWS
cat "$OUT/kbench_rows_spec_44k.txt"

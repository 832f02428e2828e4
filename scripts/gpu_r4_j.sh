#!/bin/bash
# A/B of the WIDE fused loop kernel (one launch) against the routes it replaces for SS2.0 steps at 44.1 kHz:
# k_obs_rows (no waveform buffer) / loop kernel + k_spectrogram (waveform buffer).  -DSS_AB build: SS_HIP_NO_WIDE=1 = old routing.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_j.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_j.log"
cp sound-spaces_amd/csrc/libss_hip.so /tmp/libss_hip.product.so
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_AB ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error")
trap 'cp /tmp/libss_hip.product.so "$GRAFT_REPO_ROOT/sound-spaces_amd/csrc/libss_hip.so"' EXIT
: > "$OUT/kbench_wide.txt"
for ROUND in 1 2; do
  for XF in 0 1; do
    for AG in 0 1; do
      for V in wide old; do
        if [ $V = old ]; then export SS_HIP_NO_WIDE=1; else unset SS_HIP_NO_WIDE; fi
        echo -n "$V " >> "$OUT/kbench_wide.txt"
        timeout 300 python scripts/kbench_continuous.py 128 $AG $XF 0 44100 2>/dev/null | tail -1 >> "$OUT/kbench_wide.txt"
      done
    done
  done
done
cat "$OUT/kbench_wide.txt"

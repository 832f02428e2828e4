#!/bin/bash
# Round 6: the convolution kernel ALONE against the HBM roofline, with counters instead of prose.
# For k_conv<FUSE=false> (time-domain bank) and k_conv_spec<FUSE=false> (spectral bank) at 128 / 512 / 2048 units @16 kHz and
# the 44.1 kHz loop kernels at 512 units:  (1) rocprofv3 --kernel-trace --stats  -> average launch duration
# (2) rocprofv3 --pmc FETCH_SIZE, (3) --pmc WRITE_SIZE (separate passes, no trace domains)  -> L2<->fabric bytes per launch,
# each calibrated on a known byte count in the kernel's access pattern in the same pass (scripts/calib_traffic.hip).
# scripts/conv_roofline.py turns the CSVs into profiles/r6/conv_roofline.{txt,json}.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/conv_roofline"
rm -rf "$OUT"; mkdir -p "$OUT"
run_case() {   # name, kbench args
  local NAME=$1; shift
  local CMD="python $GRAFT_REPO_ROOT/scripts/kbench.py --raw --only conv --reps 60 --distinct 8 $*"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$NAME/trace" -o t -- $CMD > "$OUT/$NAME.kbench.txt" 2>/dev/null ) || echo "trace $NAME failed"
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$NAME/fetch" -o f -- $CMD > /dev/null 2>&1 ) || echo "fetch $NAME failed"
  ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$NAME/write" -o w -- $CMD > /dev/null 2>&1 ) || echo "write $NAME failed"
  ( cd /tmp && timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/$NAME/tcc" -o c -- $CMD > /dev/null 2>&1 ) || echo "tcc $NAME failed"
}
for N in 128 512 2048; do
  run_case time16k_$N --sr 16000 --sizes $N --bank-mib 2048
  run_case spec16k_$N --sr 16000 --sizes $N --bank-mib 2048 --spectral
done
run_case time44k_512 --sr 44100 --sizes 512 --bank-mib 2048
run_case spec44k_512 --sr 44100 --sizes 512 --bank-mib 2048 --spectral
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_traffic scripts/calib_traffic.hip 2>/dev/null
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib/fetch" -o f -- /tmp/calib_traffic > /dev/null 2>&1 ) || echo "calib fetch failed"
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib/write" -o w -- /tmp/calib_traffic > /dev/null 2>&1 ) || echo "calib write failed"
python scripts/conv_roofline.py "$OUT" | tee "$OUT/conv_roofline.txt"
# keep the summaries, drop the raw CSV trees
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +

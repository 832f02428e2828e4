#!/bin/bash
# rocprofv3 kernel trace of SoundSpaces-2.0 steps at 44.1 kHz (0.25 s of a 1-s row, 128 units): plain and cross-faded -> stats_ss2_44k.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
: > "$OUT/stats_ss2_44k.txt"
for XF in 0 1; do
  D=/tmp/trace_ss2_$XF
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o trace -- python $GRAFT_REPO_ROOT/scripts/kbench_continuous.py 128 0 $XF 0 44100 2>/dev/null | tail -1 >> "$OUT/stats_ss2_44k.txt" )
  python - "$D" >> "$OUT/stats_ss2_44k.txt" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "ssk::" in row.get("Name", ""):
            print("  %-86s calls=%s avg=%.2fus min=%.2fus max=%.2fus" % (row["Name"][:86], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
done
cat "$OUT/stats_ss2_44k.txt"

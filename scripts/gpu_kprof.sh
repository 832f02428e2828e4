#!/bin/bash
# rocprofv3 kernel-trace durations of kbench variants (true kernel time, not launch-to-launch time)
# usage: gpu_kprof.sh TAG "<kbench args>" ["<kbench args 2>" ...]
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$1; shift
i=0
for ARGS in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/kprof_$TAG/$i" -o t -- python $GRAFT_REPO_ROOT/scripts/kbench.py $ARGS > /tmp/kb_$i.log 2>&1 )
  echo "== [$i] kbench $ARGS"; grep "^N=" /tmp/kb_$i.log
  python - "$GRAFT_REPO_ROOT/gpurun_out/kprof_$TAG/$i" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "ssk::" in row.get("Name", ""):
            print("   %-60s calls=%s avg=%.2fus min=%.2fus max=%.2fus" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
done

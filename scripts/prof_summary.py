#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into one text table per kernel."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    out.append(f"## {os.path.relpath(f, d)}")
    for row in csv.DictReader(open(f)):
        if "ssk::" not in row.get("Name", ""):
            continue                       # torch kernels that only build the synthetic banks
        out.append("  " + " | ".join(f"{k}={v}" for k, v in row.items()))
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if "ssk::" not in row.get("Kernel_Name", ""):
            continue
        k = row.get("Kernel_Name", "?").split("(")[0][:60]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    out.append(f"## PMC {k}  (per-dispatch mean over {max(len(v) for v in cs.values())} dispatches)")
    for c, v in sorted(cs.items()):
        out.append(f"  {c:32s} {sum(v) / len(v):16.1f}")
txt = "\n".join(out)
print(txt)
open(os.path.join(d, "summary.txt"), "w").write(txt + "\n")

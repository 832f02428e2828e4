"""Per-step GPU timeline of a rocprofv3 --kernel-trace --memory-copy-trace run: for the fused kernel's launches, the gap
between consecutive launches and what sat in it (copies, other kernels).  usage: timeline_gaps.py <rocprof out dir> [kernel substring]"""
import csv, glob, os, sys
import numpy as np

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_conv_spec<true, true>"
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K:" + r["Kernel_Name"].split("(")[0][-48:]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C:" + r.get("Direction", "copy")))
ev.sort()
idx = [i for i, e in enumerate(ev) if pat in e[2]]
print("launches of", pat, len(idx))
# split into segments of back-to-back launches (gap < 200 us)
seg, cur = [], [idx[0]]
for a, b in zip(idx, idx[1:]):
    if ev[b][0] - ev[a][1] < 200_000:
        cur.append(b)
    else:
        seg.append(cur); cur = [b]
seg.append(cur)
for s in seg:
    if len(s) < 50:
        continue
    gaps = np.array([ev[b][0] - ev[a][1] for a, b in zip(s, s[1:])]) / 1e3
    durs = np.array([ev[i][1] - ev[i][0] for i in s]) / 1e3
    period = np.array([ev[b][0] - ev[a][0] for a, b in zip(s, s[1:])]) / 1e3
    between = {}
    for a, b in zip(s, s[1:]):
        for j in range(a + 1, b):
            n = ev[j][2]
            between.setdefault(n, []).append((ev[j][1] - ev[j][0]) / 1e3)
    print("segment of %d launches: kernel %.2f us (median), gap %.2f us (median; p90 %.2f), period %.2f us" %
          (len(s), np.median(durs), np.median(gaps), np.percentile(gaps, 90), np.median(period)))
    for n, v in between.items():
        print("    in the gaps: %-60s x%d  median %.2f us" % (n, len(v), np.median(v)))

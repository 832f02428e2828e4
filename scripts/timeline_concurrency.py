"""How many launches of the fused kernel are in flight at once on the product path?  From a rocprofv3 --kernel-trace run: over
every stretch of back-to-back launches (gaps < 200 us, >= 50 launches) the start-to-start interval, the kernels' own durations
and the mean number in flight (sum of durations / span).  usage: timeline_concurrency.py <rocprof out dir> [kernel substring]"""
import csv, glob, os, sys
import numpy as np

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_conv<"
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
ev.sort()
print("launches of", pat, len(ev))
seg, cur = [], [0]
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 4):i]) < 200_000:
        cur.append(i)
    else:
        seg.append(cur); cur = [i]
seg.append(cur)
rows = []
for s in seg:
    if len(s) < 50:
        continue
    st = np.array([ev[i][0] for i in s], float); en = np.array([ev[i][1] for i in s], float)
    span = (en.max() - st.min()) / 1e3
    durs = (en - st) / 1e3
    rows.append((len(s), span / len(s), float(np.median(durs)), float(durs.sum() / span), len({ev[i][2] for i in s})))
# the longest stretches first
rows.sort(reverse=True)
print("stretch: launches | us per launch (span / n) | median kernel duration us | mean launches in flight | hardware queues seen")
for r in rows[:8]:
    print("  %5d | %6.2f | %6.2f | %4.2f | %d" % r)

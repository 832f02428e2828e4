#!/usr/bin/env python
"""rocprofv3 CSV trees of scripts/gpu_conv_roofline_r6.sh -> one table: per convolution kernel and batch size the average
launch duration (kernel trace), the ALGORITHMIC bytes (2*L*4 RIR + 2*sr*4 audiogoal per unit) and the bytes the
counters saw between L2 and the fabric (FETCH_SIZE + WRITE_SIZE, each calibrated on a known byte count in the same access
pattern), both as GB/s and as a fraction of the 8 TB/s HBM peak.  north_star: ">= 40 % HBM roofline on the FFT-convolve
kernel", "rocprof-reported achieved HBM GB/s vs peak"."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
PEAK = 8000.0
GIB = float(1 << 30)
known = {"rd8_nt": ("FETCH_SIZE", GIB), "rd16": ("FETCH_SIZE", GIB), "rd16_nt": ("FETCH_SIZE", GIB),
         "wr16_nt": ("WRITE_SIZE", GIB), "wr8_nt": ("WRITE_SIZE", GIB), "wr4": ("WRITE_SIZE", GIB / 2)}
seen = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "calib", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "").split("(")[0].strip()
        if k in known:
            seen[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
calib = {}
for k, (ctr, nbytes) in known.items():
    v = seen[k].get(ctr)
    if v:
        calib[k] = round(nbytes / (sum(v) / len(v) * 1024.0), 4)


def conv_kernel(name):
    """the FUSE=false convolution kernels of libss_hip.so (k_conv<false, ...>, k_conv_spec<false, ...>, k_conv_rows, k_conv_spec_rows)"""
    n = name.replace("void ", "").replace("ssk::", "")
    if n.startswith("k_conv_rows") or n.startswith("k_conv_spec_rows"):
        return n.split("(")[0]
    if n.startswith("k_conv<false") or n.startswith("k_conv_spec<false"):
        return n.split("(")[0]
    return None


rows = []
for d in sorted(glob.glob(os.path.join(out, "*", "trace"))):
    case = os.path.basename(os.path.dirname(d))
    form, rest = case.split("16k_") if "16k_" in case else case.split("44k_")
    sr = 16000 if "16k_" in case else 44100
    N = int(rest)
    spectral = form.startswith("spec")
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = conv_kernel(row.get("Name", ""))
            if k:
                dur[k] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
    ctr = defaultdict(lambda: defaultdict(list))
    for sub in ("fetch", "write", "tcc"):
        for f in glob.glob(os.path.join(out, case, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = conv_kernel(row.get("Kernel_Name", ""))
                if k:
                    ctr[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, (us, calls) in dur.items():
        L = sr
        alg = (2 * L * 4 + 2 * sr * 4) * N
        nblk = -(-L // 16384)
        actual_model = ((2 * nblk * 131072) if spectral else 2 * L * 4) * N + 2 * sr * 4 * N
        c = ctr.get(k, {})
        rd = "rd16_nt" if spectral else "rd8_nt"
        fetch = (sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024.0 * calib.get(rd, 1.0)) if c.get("FETCH_SIZE") else None
        write = (sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024.0 * calib.get("wr8_nt", 1.0)) if c.get("WRITE_SIZE") else None
        hit = (sum(c["TCC_HIT_sum"]) / max(1.0, sum(c["TCC_HIT_sum"]) + sum(c["TCC_MISS_sum"]))) if c.get("TCC_HIT_sum") else None
        cnt = None if fetch is None or write is None else fetch + write
        rows.append({"case": case, "kernel": k, "sr": sr, "units": N, "bank": "spectral" if spectral else "time-domain",
                     "avg_launch_us": round(us, 2), "launches": calls,
                     "algorithmic_bytes": alg, "algorithmic_GBps": round(alg / us / 1e3, 1), "algorithmic_frac": round(alg / us / 1e3 / PEAK, 4),
                     "model_actual_bytes": actual_model,
                     "counter_fetch_bytes": None if fetch is None else int(fetch), "counter_write_bytes": None if write is None else int(write),
                     "counter_GBps": None if cnt is None else round(cnt / us / 1e3, 1),
                     "counter_frac": None if cnt is None else round(cnt / us / 1e3 / PEAK, 4),
                     "tcc_hit_rate": None if hit is None else round(hit, 3)})
rows.sort(key=lambda r_: (r_["sr"], r_["bank"], r_["units"]))
print("# convolution kernel alone (audiogoal written, no STFT): rocprofv3 kernel trace + --pmc FETCH_SIZE / WRITE_SIZE (separate passes)")
print("# calibration factors (known bytes / counter bytes) of this pass: " + json.dumps(calib))
print("# algorithmic bytes = SURVEY 8(d) B_conv = (2*L*4 + 2*sr*4) per unit; counter bytes = calibrated FETCH_SIZE + WRITE_SIZE per launch")
print("%-14s %-44s %6s %10s | %12s %8s | %12s %12s %10s %8s %6s" % ("case", "kernel", "units", "avg us", "alg GB/s", "frac", "fetch MB", "write MB", "ctr GB/s", "frac", "L2hit"))
for r_ in rows:
    f = lambda v, s=1.0: "-" if v is None else ("%.1f" % (v / s))
    print("%-14s %-44s %6d %10.2f | %12.1f %8.4f | %12s %12s %10s %8s %6s" % (
        r_["case"], r_["kernel"][:44], r_["units"], r_["avg_launch_us"], r_["algorithmic_GBps"], r_["algorithmic_frac"],
        f(r_["counter_fetch_bytes"], 1e6), f(r_["counter_write_bytes"], 1e6), f(r_["counter_GBps"]),
        "-" if r_["counter_frac"] is None else "%.4f" % r_["counter_frac"], "-" if r_["tcc_hit_rate"] is None else "%.2f" % r_["tcc_hit_rate"]))
json.dump({"calibration": calib, "rows": rows}, open(os.path.join(out, "conv_roofline.json"), "w"), indent=1)

#!/bin/bash
# Round-6 profile pass (run through gpurun): for every tracked configuration
#   1. the bench line as the driver runs it (default flags of that config; the headline also at --steps 20 --warmup 5)
#   2. rocprofv3 --kernel-trace --stats of a SINGLE-STREAM run of the same config          -> stats_<cfg>.txt
# for the headline, cfg2 and cfg4: one --pmc pass per counter group (separate runs, no tracing domains mixed in) -> pmc_<cfg>.txt
# and the FETCH_SIZE / WRITE_SIZE calibration on known byte counts (scripts/calib_traffic.hip)     -> calibration in traffic.json
# -> traffic.json (keyed by the hash of the kernel sources).  Copy gpurun_out/prof_r6/* to profiles/r6/.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r6"
rm -rf "$OUT"; mkdir -p "$OUT"
declare -A CFG
CFG[headline]=""
CFG[headline_driver_protocol]="--steps 20 --warmup 5"
CFG[cfg1]="--config cfg1"
CFG[cfg2]="--config cfg2 --steps 40 --warmup 5"
CFG[cfg3]="--config cfg3"
CFG[cfg4]="--config cfg4 --steps 100"
CFG[cfg4_no_features]="--config cfg4 --steps 100 --features none"
CFG[replica44k_128]="--sr 44100 --envs 128 --steps 60 --warmup 5"
CFG[replica44k_10]="--sr 44100 --envs 10 --steps 100 --warmup 10"
CFG[replica44k_5]="--sr 44100 --envs 5 --steps 100 --warmup 10"
CFG[replica44k_10_time]="--sr 44100 --envs 10 --steps 100 --warmup 10 --rir-bank time"
CFG[cfg2_time]="--config cfg2 --steps 40 --warmup 5 --rir-bank time"
CFG[headline_time]="--rir-bank time --no-plugin-path"
CFG[cfg1_time]="--config cfg1 --rir-bank time"
CFG[cfg4_time]="--config cfg4 --steps 100 --rir-bank time"
# every line carries its cpu_baseline (the oracle at the CONFIG's own rate and shape on the box's cores, ~16 s per line)
for NAME in headline headline_driver_protocol headline_time cfg1 cfg1_time cfg2 cfg2_time cfg3 cfg4 cfg4_time cfg4_no_features replica44k_128 replica44k_10 replica44k_10_time replica44k_5; do
  ARGS=${CFG[$NAME]}
  EXTRA=""; case "$NAME" in cfg4_no_features|headline_time|cfg1_time|cfg4_time|cfg2_time|replica44k_5|replica44k_10_time) EXTRA="--no-cpu-baseline";; esac
  [ "$NAME" = headline ] || [ "$NAME" = headline_driver_protocol ] || EXTRA="$EXTRA --no-plugin-path"
  timeout 900 python bench.py $ARGS $EXTRA > "$OUT/bench_$NAME.json" 2> "$OUT/bench_$NAME.err" || echo "bench $NAME failed"
  [ "$NAME" = headline_driver_protocol ] && continue
  case "$NAME" in cfg4_no_features|cfg1_time|cfg4_time) continue;; esac
  D="$OUT/trace_$NAME"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline --no-plugin-path --no-secondary --streams 1 --regions 1 --sustain 0 > "$OUT/bench_under_rocprof_$NAME.json" 2>/dev/null )
  python - "$D" > "$OUT/stats_$NAME.txt" <<'PY'
import csv, glob, os, sys
print("# rocprofv3 --kernel-trace --stats, bench.py single-stream run (--no-secondary --streams 1): kernels of libss_hip.so")
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "ssk::" in row.get("Name", ""):
            print("%-72s calls=%s avg=%.2fus min=%.2fus max=%.2fus total=%.3fms" % (row["Name"][:72], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6))
PY
  rm -rf "$D"
done
for NAME in headline headline_time cfg1 cfg2 cfg2_time cfg4 replica44k_10; do
  ARGS=${CFG[$NAME]}
  CMD="python $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline --no-plugin-path --no-secondary --streams 1 --spinup-steps 0 --regions 1 --sustain 0"
  D="$OUT/pmc_$NAME"
  i=0
  for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --pmc $PMC --output-format csv -d "$D" -o pmc$i -- $CMD > /dev/null 2>&1 ) || echo "pmc pass $i of $NAME failed"
  done
  python scripts/prof_summary.py "$D" > /dev/null 2>&1
  cp "$D/summary.txt" "$OUT/pmc_$NAME.txt"
done
# ---- calibration of the two traffic counters on known byte counts, in this library's access patterns
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_traffic scripts/calib_traffic.hip 2>/dev/null
D="$OUT/pmc_calib"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D" -o fetch -- /tmp/calib_traffic > /dev/null 2>&1 ) || echo "calib fetch pass failed"
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D" -o write -- /tmp/calib_traffic > /dev/null 2>&1 ) || echo "calib write pass failed"
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
# calibration: counter (KiB) per launch of each pattern kernel against its known bytes
GIB = float(1 << 30)
known = {"rd8_nt": ("FETCH_SIZE", GIB), "rd16": ("FETCH_SIZE", GIB), "rd16_nt": ("FETCH_SIZE", GIB),
         "wr16_nt": ("WRITE_SIZE", GIB), "wr8_nt": ("WRITE_SIZE", GIB), "wr4": ("WRITE_SIZE", GIB / 2)}
seen = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_calib", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "").split("(")[0].strip()
        if k in known:
            seen[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
calib = {}
for k, (ctr, nbytes) in known.items():
    v = seen[k].get(ctr)
    if v:
        rep = sum(v) / len(v) * 1024.0
        calib[k] = {"counter": ctr, "known_bytes": nbytes, "reported_bytes": rep, "factor": round(nbytes / rep, 4)}
    other = "WRITE_SIZE" if ctr == "FETCH_SIZE" else "FETCH_SIZE"
    v2 = seen[k].get(other)
    if v2 and k in calib:
        calib[k]["other_counter_bytes"] = sum(v2) / len(v2) * 1024.0      # (a write kernel that also fetches: read-for-ownership)
class Names(dict):                       # rocprofv3's template spelling -> bench.py's kernel names
    def _name(self, k):
        for kk in ("k_obs_rows", "k_obs_blocks"):
            if k.startswith(kk + "<"):
                return kk + ("<SPECTRAL=true>" if k.startswith(kk + "<true") else "<SPECTRAL=false>")
        if k.startswith("k_conv_spec<"):     # <FUSE, SIMPLE, TAB>
            b = [x.strip() == "true" for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
            return "k_conv_spec<FUSE=%s%s>" % ("true" if b[0] else "false", "" if b[1] else ",loop")
        if k.startswith("k_conv<"):          # <FUSE, SIMPLE, XFADE, TAB, WIDE>
            b = [x.strip() == "true" for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
            if len(b) >= 3 and not b[2]:
                return "k_conv<FUSE=%s%s>" % ("true" if b[0] else "false", "" if b[1] else ",loop")
        if k.startswith("k_features<"):      # <MEL, SG, GCC>
            b = [x.strip() == "true" for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
            return "k_features<" + ",".join(n for n, on in zip(("spectrogram", "logmel", "gccphat"), (b[1], b[0], b[2])) if on) + ">"
        return None
    def __contains__(self, k):
        return self._name(k) is not None
    def __getitem__(self, k):
        return self._name(k)
names = Names()
# the pattern that carries (nearly) all of a kernel's bytes on each side: RIR rows are 8-B/lane nt loads; the stash of k_obs_rows
# goes out as 16-B nt stores and comes back as 16-B loads next to 8-B row loads (fetch side mixed: both factors are given)
dominant = {"k_conv<FUSE=true>": ("rd8_nt", "wr4"), "k_conv<FUSE=true,loop>": ("rd8_nt", "wr8_nt"),
            "k_conv<FUSE=false,loop>": ("rd8_nt", "wr8_nt"), "k_conv<FUSE=false>": ("rd8_nt", "wr8_nt"),
            "k_obs_rows<SPECTRAL=false>": ("rd16", "wr16_nt"), "k_obs_rows<SPECTRAL=true>": ("rd16_nt", "wr4"),
            "k_obs_blocks<SPECTRAL=false>": ("rd8_nt", "wr4"), "k_obs_blocks<SPECTRAL=true>": ("rd16_nt", "wr4"),
            "k_conv_spec<FUSE=true>": ("rd16_nt", "wr4"), "k_conv_spec<FUSE=true,loop>": ("rd16_nt", "wr8_nt"),
            "k_conv_spec<FUSE=false,loop>": ("rd16_nt", "wr8_nt"), "k_conv_spec<FUSE=false>": ("rd16_nt", "wr8_nt"),
            "k_features<logmel,gccphat>": ("rd16", "wr4"),
            "k_features<spectrogram,logmel,gccphat>": ("rd16", "wr4")}
kernels = {}
for cfg in ("headline", "headline_time", "cfg1", "cfg2", "cfg2_time", "cfg4", "replica44k_10"):
    d = os.path.join(out, "pmc_" + cfg)
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "ssk::" in row.get("Kernel_Name", ""):
                agg[row["Kernel_Name"].split("(")[0].replace("void ssk::", "").replace("ssk::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    try:
        bench = json.loads(open(os.path.join(out, "bench_under_rocprof_%s.json" % cfg)).read().strip().splitlines()[-1])
    except Exception:
        continue
    def is_tab(k):                       # k_conv<FUSE, SIMPLE, XFADE, TAB, WIDE>: the unit-table instantiation is the product path's
        if not k.startswith("k_conv<"):
            return False
        b = [x.strip() == "true" for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
        return len(b) >= 4 and b[3]
    for k, cs in sorted(agg.items(), key=lambda kv: is_tab(kv[0])):      # ... so it is read LAST and is the one recorded
        if k in names and names[k] in dominant and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            nm = names[k]
            f_kib = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]); w_kib = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            rd, wr = dominant[nm]
            fc = calib.get(rd, {}).get("factor", 1.0); wc = calib.get(wr, {}).get("factor", 1.0)
            if wr == "wr4":
                wc = 1.0        # WRITE_SIZE counts the bytes that MOVE: 4-byte stores to every other float of a line are written
                                # as whole sectors (calibration: 2x the bytes stored) - that IS the traffic; the factor 0.5 would
                                # turn it back into bytes stored
            kernels["%s@%d" % (nm, bench["config"]["units_per_gpu"])] = {
                "units_per_launch": bench["config"]["units_per_gpu"], "sampling_rate": bench["config"]["sampling_rate"],
                "fetch_bytes": f_kib * 1024, "write_bytes": w_kib * 1024,
                "fetch_correction": fc, "write_correction": wc,
                "fetch_pattern": rd, "write_pattern": wr,
                "tcc_hit_rate": round(sum(cs["TCC_HIT_sum"]) / (sum(cs["TCC_HIT_sum"]) + sum(cs["TCC_MISS_sum"])), 3) if "TCC_HIT_sum" in cs else None,
                "note": "per-dispatch means of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB x 1024; separate passes), each multiplied by the "
                        "factor measured in the SAME pass script on a known byte count in the kernel's dominant access pattern "
                        "(scripts/calib_traffic.hip; factors under 'calibration')"}
src_hash = open("sound-spaces_amd/csrc/.libss_hip.kernelhash").read().strip()     # device code only (build.py::kernel_hash)
json.dump({"source_hash": src_hash, "command": "bench.py [--config cfg2|cfg4] --no-cpu-baseline --no-plugin-path --no-secondary --streams 1 --spinup-steps 0 --regions 1 --sustain 0",
           "calibration": calib, "kernels": kernels}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps({"calibration": calib, "kernels": kernels}, indent=1))
PY
rm -rf "$OUT"/pmc_headline "$OUT"/pmc_headline_time "$OUT"/pmc_cfg1 "$OUT"/pmc_cfg2 "$OUT"/pmc_cfg2_time "$OUT"/pmc_cfg4 "$OUT"/pmc_replica44k_10 "$OUT"/pmc_calib
for f in "$OUT"/stats_*.txt; do echo "== $f"; cat "$f"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prof_r6/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value',d['value'], 'ms',d['ms_per_step'], 'roofline',d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'pipe',d['roofline'].get('pipeline_frac'), {k:v.get('value') for k,v in d.items() if isinstance(v,dict) and 'value' in v and k not in ('roofline','cpu_baseline')})
    except Exception as e:
        print(f,'ERR',e)
PY

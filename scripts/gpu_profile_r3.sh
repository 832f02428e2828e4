#!/bin/bash
# Round-3 profile pass (run through gpurun): for every tracked configuration
#   1. the bench line as the driver runs it (default flags of that config)               -> bench_<cfg>.json
#   2. rocprofv3 --kernel-trace --stats of a SINGLE-STREAM run of the same config          -> stats_<cfg>.txt
# and for the headline and cfg2: one --pmc pass per counter group (separate runs, no tracing domains mixed in)
# -> pmc_<cfg>.txt, traffic.json (keyed by the hash of the kernel sources).  Copy gpurun_out/prof_r3/* to profiles/r3/.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r3"
mkdir -p "$OUT"
declare -A CFG
CFG[headline]=""
CFG[cfg1]="--config cfg1"
CFG[cfg2]="--config cfg2 --steps 40 --warmup 5"
CFG[cfg4]="--config cfg4 --steps 100"
CFG[replica44k_128]="--sr 44100 --envs 128 --steps 60 --warmup 5"
for NAME in headline cfg1 cfg2 cfg4 replica44k_128; do
  ARGS=${CFG[$NAME]}
  EXTRA="--no-cpu-baseline"; [ "$NAME" = headline ] && EXTRA=""
  timeout 900 python bench.py $ARGS $EXTRA > "$OUT/bench_$NAME.json" 2> "$OUT/bench_$NAME.err" || echo "bench $NAME failed"
  D="$OUT/trace_$NAME"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline --no-plugin-path --no-secondary --streams 1 > "$OUT/bench_under_rocprof_$NAME.json" 2>/dev/null )
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
for NAME in headline cfg2; do
  ARGS=${CFG[$NAME]}
  CMD="python $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline --no-plugin-path --no-secondary --streams 1 --spinup-steps 0"
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
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
names = {"k_conv<true, true, false, false>": "k_conv<FUSE=true>", "k_conv<true, true, false, true>": "k_conv<FUSE=true>",
         "k_obs_rows<false>": "k_obs_rows<SPECTRAL=false>", "k_obs_rows<false, false>": "k_obs_rows<SPECTRAL=false>"}        # (the TAB instantiation moves the same bytes)
kernels = {}
for cfg in ("headline", "cfg2"):
    d = os.path.join(out, "pmc_" + cfg)
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "ssk::" in row.get("Kernel_Name", ""):
                agg[row["Kernel_Name"].split("(")[0].replace("void ssk::", "").replace("ssk::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    bench = json.loads(open(os.path.join(out, "bench_under_rocprof_%s.json" % cfg)).read().strip().splitlines()[-1])
    for k, cs in agg.items():
        if k in names and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f_kib = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]); w_kib = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            kernels[names[k]] = {
                "units_per_launch": bench["config"]["units_per_gpu"], "sampling_rate": bench["config"]["sampling_rate"],
                "fetch_bytes": f_kib * 1024, "write_bytes": w_kib * 1024,
                "tcc_hit_rate": round(sum(cs["TCC_HIT_sum"]) / (sum(cs["TCC_HIT_sum"]) + sum(cs["TCC_MISS_sum"])), 3) if "TCC_HIT_sum" in cs else None,
                "note": "per-dispatch means of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB x 1024; separate passes).  FETCH_SIZE is the raw "
                        "counter: on gfx950 it tallies 64 B per 128-B request of a wide (16 B/lane) streaming read (MI355X_MICROARCH: x2 for "
                        "such streams)",
                "fetch_correction": 1.0,
                "correction_note": "mixed widths (RIR rows 8 B/lane, window / block spectra 16 B/lane): FETCH_SIZE left raw (uncalibrated "
                                   "for 8-B/lane loads; an upper bound of the true figure is 2x)"}
src_hash = open("sound-spaces_amd/csrc/.libss_hip.srchash").read().strip()
json.dump({"source_hash": src_hash, "command": "bench.py [--config cfg2] --no-cpu-baseline --no-plugin-path --no-secondary --streams 1 --spinup-steps 0",
           "kernels": kernels}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(kernels, indent=1))
PY
rm -rf "$OUT"/pmc_headline "$OUT"/pmc_cfg2
for f in "$OUT"/stats_*.txt; do echo "== $f"; cat "$f"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prof_r3/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value',d['value'], 'ms',d['ms_per_step'], 'roofline',d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'pipe',d['roofline'].get('pipeline_frac'), {k:v.get('value') for k,v in d.items() if isinstance(v,dict) and 'value' in v and k not in ('roofline','cpu_baseline')})
    except Exception as e:
        print(f,'ERR',e)
PY

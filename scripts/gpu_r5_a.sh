#!/bin/bash
# Round 5, pass A: GPU test-suite on the new build; same-box A/B of the split rows (ConvParams::parts_log2) with the -DSS_AB
# library prebuilt in gpurun_in/ (SS_HIP_PARTS_LOG2=0 = one workgroup per row, unset = the product's choice); the RIR miss
# path (scripts/bench_loader.py); SS2.0 deferred mode; cfg1 / cfg3 lines with a rocprofv3 kernel trace.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r5a"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_gpu.log"
# ---- A/B: split rows
cp sound-spaces_amd/csrc/libss_hip.so /tmp/libss_hip.product.so
cp gpurun_in/libss_hip_ab.so sound-spaces_amd/csrc/libss_hip.so
for rep in 1 2; do
  for V in 0 auto; do
    if [ $V = auto ]; then unset SS_HIP_PARTS_LOG2; else export SS_HIP_PARTS_LOG2=$V; fi
    echo "== parts=$V rep=$rep time-domain" >> "$OUT/kbench_parts.txt"
    timeout 300 python scripts/kbench.py --raw --only fused --sizes 1,2,4,8,16,32,48,64,128 --reps 400 --bank-mib 256 >> "$OUT/kbench_parts.txt" 2>&1
  done
done
for V in 0 auto; do
  if [ $V = auto ]; then unset SS_HIP_PARTS_LOG2; else export SS_HIP_PARTS_LOG2=$V; fi
  echo "== parts=$V spectral" >> "$OUT/kbench_parts.txt"
  timeout 300 python scripts/kbench.py --raw --only fused --spectral --sizes 1,8,16,32,64 --reps 400 --bank-mib 256 >> "$OUT/kbench_parts.txt" 2>&1
done
for V in 1 2 3; do
  export SS_HIP_PARTS_LOG2=$V
  echo "== parts=$V forced time-domain" >> "$OUT/kbench_parts.txt"
  timeout 300 python scripts/kbench.py --raw --only fused --sizes 1,8,16,32 --reps 400 --bank-mib 256 >> "$OUT/kbench_parts.txt" 2>&1
done
unset SS_HIP_PARTS_LOG2
cp /tmp/libss_hip.product.so sound-spaces_amd/csrc/libss_hip.so
cat "$OUT/kbench_parts.txt"
# ---- loader / miss path
timeout 600 python scripts/bench_loader.py --out "$OUT/loader.json" > "$OUT/loader.log" 2>&1; echo "loader rc=$?"; tail -25 "$OUT/loader.log"
# ---- SS2.0 deferred
timeout 300 python scripts/bench_deferred_continuous.py > "$OUT/bench_deferred_continuous.json" 2> "$OUT/bench_deferred_continuous.err"; echo "cont rc=$?"; cat "$OUT/bench_deferred_continuous.json"; tail -3 "$OUT/bench_deferred_continuous.err"
# ---- small-step config lines + kernel trace
for C in cfg1 cfg3; do
  timeout 600 python bench.py --config $C --no-plugin-path > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"; echo "$C rc=$?"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$C -o run -- python "$GRAFT_REPO_ROOT/bench.py" --config $C --no-plugin-path --no-cpu-baseline --no-secondary --sustain 0 --streams 1 --regions 1 > "$OUT/bench_under_rocprof_$C.json" 2> /tmp/prof_$C.err)
  python scripts/prof_summary.py /tmp/prof_$C > "$OUT/stats_$C.txt" 2>&1 || find /tmp/prof_$C -name "*kernel_stats*" | head -1 | xargs -I{} cp {} "$OUT/stats_$C.csv"
  head -12 "$OUT/stats_$C.txt"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5a/bench_cfg*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1],'value',d['value'],'ms',d['ms_per_step'],'roofline',d['roofline']['frac'],d['roofline'].get('avg_launch_ms'),'cpu',d.get('cpu_baseline',{}).get('value'))
    except Exception as e:
        print(f,'ERR',e)
PY

#!/bin/bash
# Round 5, pass B: tests; A/B of the sorted unit table (-DSS_AB library prebuilt in gpurun_in/: SS_HIP_NO_SORT=1 = the caller's
# unit order) on the headline; the miss path after the C-retry rework; SS2.0 deferred; small-step lines with the share-aware
# split; k_features timing + LDS conflict counters; eager profile.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r5b"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_gpu.log"
cp sound-spaces_amd/csrc/libss_hip.so /tmp/libss_hip.product.so
cp gpurun_in/libss_hip_ab.so sound-spaces_amd/csrc/libss_hip.so
for rep in 1 2 3; do
  for V in sort nosort; do
    if [ $V = nosort ]; then export SS_HIP_NO_SORT=1; else unset SS_HIP_NO_SORT; fi
    timeout 300 python bench.py --no-cpu-baseline --no-plugin-path --sustain 0 > "$OUT/ab_${V}_$rep.json" 2>/dev/null
    python - "$OUT/ab_${V}_$rep.json" $V <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2],'value',round(d['value']/1e6,3),'ms',d['ms_per_step'],'ctx_single',d['ctx_single_stream']['ms_per_step'],'preplanned',d['preplanned_single_stream']['ms_per_step'],'kernel',d['roofline']['avg_launch_ms'],'spectral',round(d['spectral_bank']['value']/1e6,3))
PY
  done
done
unset SS_HIP_NO_SORT
# TCC hit rate / traffic of the headline kernel, sorted vs not
for V in sort nosort; do
  if [ $V = nosort ]; then export SS_HIP_NO_SORT=1; else unset SS_HIP_NO_SORT; fi
  i=0
  for PMC in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" ; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_$V -o p$i -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-plugin-path --no-secondary --spinup-steps 0 --regions 1 --sustain 0 --steps 60 > /dev/null 2>&1 ) || echo "pmc $V $i failed"
  done
  python scripts/prof_summary.py /tmp/pmc_$V > /dev/null 2>&1; cp /tmp/pmc_$V/summary.txt "$OUT/pmc_headline_$V.txt"; grep -A4 "k_conv<true, true, false, true" "$OUT/pmc_headline_$V.txt" | head -8
done
unset SS_HIP_NO_SORT
cp /tmp/libss_hip.product.so sound-spaces_amd/csrc/libss_hip.so
# ---- loader / miss path, SS2.0
timeout 600 python scripts/bench_loader.py --out "$OUT/loader.json" > "$OUT/loader.log" 2>&1; echo "loader rc=$?"; grep miss_rate "$OUT/loader.log" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['mode'],d['reader'],d['miss_rate'],d['trainer_half_us_per_step_median'],d['env_steps_per_s_trainer_half'])"
timeout 300 python scripts/bench_deferred_continuous.py > "$OUT/bench_deferred_continuous.json" 2> "$OUT/bench_deferred_continuous.err"; cat "$OUT/bench_deferred_continuous.json"
timeout 300 python scripts/bench_deferred_continuous.py --profile > "$OUT/bench_deferred_continuous_profile.txt" 2>&1; sed -n 1,30p "$OUT/bench_deferred_continuous_profile.txt"
# ---- small steps: product path
for C in cfg1 cfg3; do
  timeout 600 python bench.py --config $C --no-plugin-path --no-cpu-baseline > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"; echo "$C rc=$?"
done
# ---- k_features
timeout 300 python scripts/kbench_features.py > "$OUT/kbench_features.json" 2>/dev/null; cat "$OUT/kbench_features.json"
CMD="python $GRAFT_REPO_ROOT/bench.py --config cfg4 --steps 40 --no-cpu-baseline --no-plugin-path --no-secondary --streams 1 --spinup-steps 0 --regions 1 --sustain 0"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_cfg4 -o p1 -- $CMD > /dev/null 2>&1 ) || echo "pmc cfg4 failed"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d /tmp/pmc_cfg4 -o p2 -- $CMD > /dev/null 2>&1 ) || echo "pmc cfg4 failed"
python scripts/prof_summary.py /tmp/pmc_cfg4 > /dev/null 2>&1; cp /tmp/pmc_cfg4/summary.txt "$OUT/pmc_cfg4.txt"; cat "$OUT/pmc_cfg4.txt" | head -40
timeout 300 python scripts/prof_eager.py > "$OUT/prof_eager.txt" 2>&1; grep "^eager" "$OUT/prof_eager.txt"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5b/bench_cfg*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1],'value',d['value'],'ms',d['ms_per_step'],'ctx_single',d['ctx_single_stream']['ms_per_step'],'kernel',d['roofline'].get('avg_launch_ms'))
    except Exception as e:
        print(f,'ERR',e)
PY

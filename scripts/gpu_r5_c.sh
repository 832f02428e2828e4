#!/bin/bash
# Round 5, pass C: tests; do two lanes overlap at small steps (scripts/kbench_lanes.py, forced parts); k_obs_rows split rows
# at the reference's Replica rate with few envs (A/B); product-path lines for the small configurations.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r5c"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_gpu.log"
cp sound-spaces_amd/csrc/libss_hip.so /tmp/libss_hip.product.so
cp gpurun_in/libss_hip_ab.so sound-spaces_amd/csrc/libss_hip.so
for V in 0 1 2 3; do
  export SS_HIP_PARTS_LOG2=$V
  timeout 300 python scripts/kbench_lanes.py --sizes 8,16,32,64 >> "$OUT/kbench_lanes.txt" 2>/dev/null
done
unset SS_HIP_PARTS_LOG2
timeout 300 python scripts/kbench_lanes.py --sizes 8,16,32,64,128 >> "$OUT/kbench_lanes.txt" 2>/dev/null
cat "$OUT/kbench_lanes.txt"
for rep in 1 2; do
  for V in 0 auto; do
    if [ $V = auto ]; then unset SS_HIP_PARTS_LOG2; else export SS_HIP_PARTS_LOG2=$V; fi
    echo "== 44.1 kHz parts=$V rep=$rep" >> "$OUT/kbench_parts_44k.txt"
    timeout 300 python scripts/kbench.py --sr 44100 --raw --only fused --sizes 1,2,5,10,16,32,64,128 --reps 200 --bank-mib 512 >> "$OUT/kbench_parts_44k.txt" 2>/dev/null
  done
done
echo "== 44.1 kHz spectral parts=0" >> "$OUT/kbench_parts_44k.txt"; export SS_HIP_PARTS_LOG2=0
timeout 300 python scripts/kbench.py --sr 44100 --raw --only fused --spectral --sizes 1,5,10,32 --reps 200 --bank-mib 512 >> "$OUT/kbench_parts_44k.txt" 2>/dev/null
echo "== 44.1 kHz spectral parts=auto" >> "$OUT/kbench_parts_44k.txt"; unset SS_HIP_PARTS_LOG2
timeout 300 python scripts/kbench.py --sr 44100 --raw --only fused --spectral --sizes 1,5,10,32 --reps 200 --bank-mib 512 >> "$OUT/kbench_parts_44k.txt" 2>/dev/null
cat "$OUT/kbench_parts_44k.txt"
cp /tmp/libss_hip.product.so sound-spaces_amd/csrc/libss_hip.so
for C in cfg1 cfg3; do
  timeout 600 python bench.py --config $C --no-plugin-path --no-cpu-baseline > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"; echo "$C rc=$?"
done
timeout 600 python bench.py --sr 44100 --envs 10 --steps 100 --no-plugin-path --no-cpu-baseline > "$OUT/bench_replica44k_10.json" 2> "$OUT/bench_replica44k_10.err"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1],'value',d['value'],'ms',d['ms_per_step'],'ctx_single',d['ctx_single_stream']['ms_per_step'],'kernel',d['roofline'].get('avg_launch_ms'),'gpu_ms',d.get('gpu_ms_per_step',{}).get('median'))
    except Exception as e:
        print(f,'ERR',e)
PY

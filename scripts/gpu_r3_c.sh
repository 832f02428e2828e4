#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r3c"; mkdir -p "$OUT"
bash scripts/gpu_rows_ladder.sh "--sr 44100 --sizes 128,512 --raw --only fused --reps 100 --bank-mib 1024" 32 64 128 256 288 384 > "$OUT/variants_time.txt" 2>&1
echo "xcd_map=0: $(SS_HIP_XCD_MAP=0 timeout 300 python scripts/kbench.py --sr 44100 --sizes 128,512 --raw --only fused --reps 100 --bank-mib 1024 2>&1 | grep '^N=' | tr '\n' ' ')" >> "$OUT/variants_time.txt"
cat "$OUT/variants_time.txt"

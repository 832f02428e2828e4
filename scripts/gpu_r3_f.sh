#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r3f"; mkdir -p "$OUT"
bash scripts/gpu_rows_ladder.sh "--sr 44100 --sizes 128,512 --raw --only fused --reps 100 --bank-mib 1024" 512 1024 1536 > "$OUT/variants2_time.txt" 2>&1
cat "$OUT/variants2_time.txt"

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r3g"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -x -q -k "fused_rows or 44k or bucketed or config2 or spectral" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
bash scripts/gpu_rows_ladder.sh "--sr 44100 --sizes 128,512 --raw --only fused --reps 100 --bank-mib 1024" 1 2 4 8 16 > "$OUT/ladder_time.txt" 2>&1
cat "$OUT/ladder_time.txt"
echo "spectral: $(timeout 300 python scripts/kbench.py --sr 44100 --sizes 128,512 --raw --only fused --reps 100 --bank-mib 1024 --spectral 2>&1 | grep '^N=' | tr '\n' ' ')"

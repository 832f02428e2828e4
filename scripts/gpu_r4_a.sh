#!/bin/bash
# round 4, first GPU pass: A/B of the two FFT cores (kbench32), then the GPU test-suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
timeout 600 python scripts/kbench32.py --out "$OUT/kbench32_a.json" > "$OUT/kbench32_a.log" 2>&1; echo "kbench32 rc=$?"; tail -12 "$OUT/kbench32_a.log"
timeout 1500 python -m pytest tests -m gpu -q -x > "$OUT/pytest_a.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_a.log"

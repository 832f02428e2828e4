#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
timeout 600 python scripts/bench_deferred_continuous.py > "$OUT/bench_deferred_continuous.json" 2> "$OUT/bench_deferred_continuous.err"; echo "rc=$?"; cat "$OUT/bench_deferred_continuous.json"; tail -3 "$OUT/bench_deferred_continuous.err"
timeout 900 python -m pytest tests -m gpu -q -x -k "continuous or deferred or live or plugin or adapter or store or bucket" > "$OUT/pytest_h.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_h.log"

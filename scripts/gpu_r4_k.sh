#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_k.log" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/pytest_k.log"
timeout 600 python scripts/bench_boundary.py > "$OUT/bench_boundary_k.jsonl" 2> "$OUT/bench_boundary_k.err"; echo "boundary rc=$?"; cat "$OUT/bench_boundary_k.jsonl"; tail -3 "$OUT/bench_boundary_k.err"

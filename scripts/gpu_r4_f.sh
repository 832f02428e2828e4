#!/bin/bash
# round 4, GPU pass F: full GPU test-suite with the TORCH_LIBRARY extension loaded, the eager boundary mode, all boundary modes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4"; mkdir -p "$OUT"
python -c "import sys; sys.path[:0]=['sound-spaces_amd']; from ss_amd import ops; print('native ops:', ops.NATIVE_OPS)"
timeout 1500 python -m pytest tests -m gpu -q -x > "$OUT/pytest_f.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest_f.log"
timeout 300 python scripts/prof_eager.py > "$OUT/prof_eager_f.txt" 2>&1; echo "prof rc=$?"; head -40 "$OUT/prof_eager_f.txt"
timeout 600 python scripts/bench_boundary.py > "$OUT/bench_boundary_f.jsonl" 2> "$OUT/bench_boundary_f.err"; echo "boundary rc=$?"; cat "$OUT/bench_boundary_f.jsonl"

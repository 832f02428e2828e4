#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/tests"; mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -25 "$OUT/pytest.log"

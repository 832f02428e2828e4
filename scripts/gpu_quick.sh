#!/bin/bash
# quick check: parity tests + per-kernel timings (two runs)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout 200 python scripts/kbench.py --sizes 128,2048 --reps 100 2>&1 | grep "^N=" | tr '\n' ' '; echo; done

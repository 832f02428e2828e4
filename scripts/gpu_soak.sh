#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2 3; do timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -1; done
for s in 1 2 3; do timeout 900 python scripts/gpu_soak.py $s 30 2>&1 | tail -2; done

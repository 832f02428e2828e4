#!/bin/bash
# A/B: persistent row kernel (default) vs one workgroup per row (SS_HIP_NO_ROW_KERNEL=1)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
  echo "rows kernel:   $(timeout 200 python scripts/kbench.py --sizes 128,512,2048 --reps 100 2>&1 | grep '^N=' | tr '\n' ' ')"
  echo "one WG per row: $(SS_HIP_NO_ROW_KERNEL=1 timeout 200 python scripts/kbench.py --sizes 128,512,2048 --reps 100 2>&1 | grep '^N=' | tr '\n' ' ')"
done

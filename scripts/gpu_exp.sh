#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for B in 512 8; do echo "== bank ${B} MiB"; for i in 1 2; do timeout 200 python scripts/kbench.py --sizes 128,2048 --reps 100 --bank-mib $B 2>&1 | grep "^N=" | tr '\n' ' '; echo; done; done

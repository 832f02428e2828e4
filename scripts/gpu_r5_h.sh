#!/bin/bash
# pass H: soak of the overlap mode with split rows at several step sizes (bit-identical to the single-stream context)
cd "$GRAFT_REPO_ROOT" || exit 1
for N in 3 16 32 64; do timeout 600 python scripts/soak_ctx.py 120 $N 2>&1 | tail -1; done

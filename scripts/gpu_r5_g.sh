#!/bin/bash
# pass G: smoke() + the final pass
cd "$GRAFT_REPO_ROOT" || exit 1
python __graft_entry__.py smoke 2>&1 | tail -2
bash scripts/gpu_final_r5.sh

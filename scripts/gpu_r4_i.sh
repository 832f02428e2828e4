#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python scripts/bench_deferred_continuous.py --profile 2>&1 | grep -v amdgpu.ids | head -60

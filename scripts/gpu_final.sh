#!/bin/bash
# round-end style pass: smoke, the whole GPU test suite, then the profile pass
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/final/pytest.log
bash scripts/gpu_profile_r3.sh > gpurun_out/final/profile.log 2>&1; tail -12 gpurun_out/final/profile.log

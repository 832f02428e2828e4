#!/bin/bash
# pass D: trial of the final pass + a cProfile of deferred-mode steps with 5 % pose misses
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_final_r5.sh
timeout 300 python scripts/bench_loader.py --profile-miss 0.05 --steps 60 > gpurun_out/prof_r5/miss_profile.txt 2>&1; head -60 gpurun_out/prof_r5/miss_profile.txt

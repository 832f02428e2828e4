#!/bin/bash
# like gpu_ab_files.sh for two files at once: cur = tree as sent, alt = both files replaced
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
C=sound-spaces_amd/csrc
cp $C/ss_kernels.hpp /tmp/k.cur; cp $C/ss_fft_core.hpp /tmp/c.cur
for round in 1 2; do
  for V in cur alt; do
    if [ $V = alt ]; then cp gpurun_in/ss_kernels.old.hpp $C/ss_kernels.hpp; cp gpurun_in/ss_fft_core.old.hpp $C/ss_fft_core.hpp; else cp /tmp/k.cur $C/ss_kernels.hpp; cp /tmp/c.cur $C/ss_fft_core.hpp; fi
    (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error")
    echo "$V: $(timeout 200 python scripts/kbench.py ${KB_ARGS:---sizes 128,2048 --reps 200} 2>&1 | grep '^N=' | tr '\n' ' ')"
  done
done

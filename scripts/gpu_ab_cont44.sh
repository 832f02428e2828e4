#!/bin/bash
# SS2.0 steps (0.25 s of a 1-s row), kernel level (scripts/kbench_continuous.py): the in-tree library against the previous
# commit's (gpurun_in/libss_hip_old.so; see gpu_ab_so.sh for how to build it), 16 kHz and 44.1 kHz, plain and cross-faded
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
SO=sound-spaces_amd/csrc/libss_hip.so
cp $SO /tmp/new.so
for V in new old new old; do
  cp /tmp/new.so $SO; [ $V = old ] && cp gpurun_in/libss_hip_old.so $SO
  for SR in 16000 44100; do for X in 0 1; do
    echo -n "$V: "; timeout 300 python scripts/kbench_continuous.py 128 1 $X 0 $SR 2>/dev/null | tail -1
  done; done
done
cp /tmp/new.so $SO

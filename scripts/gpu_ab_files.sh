#!/bin/bash
# A/B of two versions of a source file on the same box: gpu_ab_files.sh <repo-relative file> <alternative copy under gpurun_in/>
# (KB_ARGS overrides the kbench arguments, e.g. KB_ARGS='--sr 44100 --sizes 128,512 --reps 50 --bank-mib 768')
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
F=$1; ALT=$2
cp "$F" /tmp/cur.src
for round in 1 2; do
  for V in cur alt; do
    if [ $V = alt ]; then cp "$ALT" "$F"; else cp /tmp/cur.src "$F"; fi
    (cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error")
    echo "$V: $(timeout 200 python scripts/kbench.py ${KB_ARGS:---sizes 128,2048 --reps 200} 2>&1 | grep '^N=' | tr '\n' ' ')"
  done
done
cp /tmp/cur.src "$F"

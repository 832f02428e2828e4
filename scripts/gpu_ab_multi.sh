#!/bin/bash
# Several versions of one source file on the same box: gpu_ab_multi.sh <repo-relative file> <alt1> <alt2> ...  (copies under
# gpurun_in/); the current file is measured first and last.  KB_ARGS overrides the kbench arguments.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
F=$1; shift
cp "$F" /tmp/cur.src
build() { (cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error"); }
run() { echo "$1: $(timeout 200 python scripts/kbench.py ${KB_ARGS:---sizes 128 --reps 300} 2>&1 | grep '^N=' | sed 's/.*dbg=[0-9]* //' | tr '\n' ' ')"; }
build; run cur
for ALT in "$@"; do cp "$ALT" "$F"; build; run "$(basename $ALT)"; done
cp /tmp/cur.src "$F"; build; run cur

#!/bin/bash
# A/B of build variants on the GPU box: rebuild libss_hip.so with extra -D flags (one variant per argument; "" = base),
# time the kernels each time.  usage: gpu_ab.sh [--only spec] -- "" "-DFOO" "-DFOO -DBAR"
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
ONLY=""; if [ "${1:-}" = "--only" ]; then ONLY="--only $2"; shift 2; fi
[ "${1:-}" = "--" ] && shift
cp sound-spaces_amd/csrc/libss_hip.so /tmp/base.so
for V in "$@"; do
  (cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $V ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error" )
  echo "== variant [$V]"
  for i in 1 2; do timeout 200 python scripts/kbench.py --sizes 128,2048 --reps 100 $ONLY 2>&1 | grep "^N=" | tr '\n' ' '; echo; done
done
cp /tmp/base.so sound-spaces_amd/csrc/libss_hip.so

#!/bin/bash
# k_obs_rows timing ablations on ONE box: builds libss_hip.so with -DSS_ROWS_ABL=<mask> for every mask given (results are
# wrong by construction, only the time is read) and runs kbench at 44.1 kHz; the product build is measured first and last.
# usage: gpu_rows_ladder.sh "<kbench args>" mask1 mask2 ...
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
KB_ARGS=${1:---sr 44100 --sizes 128,512 --raw --only fused --reps 100 --bank-mib 1024}; shift
build() { (cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $1 ss_hip.hip -o libss_hip.so 2>&1 | grep -E "error"); }
run() { echo "$1: $(timeout 300 python scripts/kbench.py $KB_ARGS 2>&1 | grep '^N=' | sed 's/ raw//; s/map=[0-9]* sort=[0-9]* dbg=[0-9]* //' | tr '\n' ' ')"; }
cp sound-spaces_amd/csrc/libss_hip.so /tmp/product.so
run product
for M in "$@"; do build "-DSS_ROWS_ABL=$M"; run "abl=$M"; done
cp /tmp/product.so sound-spaces_amd/csrc/libss_hip.so
run product

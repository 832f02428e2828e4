#!/bin/bash
# build it here first: in csrc/ss_fft_core.hpp::lds_barrier() drop "s_barrier" from the asm string, then
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude sound-spaces_amd/csrc/ss_hip.hip -o gpurun_in/libss_hip_nobar.so
# and restore the header (git checkout).  NEVER ship that library: its results are wrong.
# TIMING ONLY (results are wrong by construction): the in-tree library against a build whose workgroup barriers are reduced
# to the wave's own s_waitcnt lgkmcnt(0) (gpurun_in/libss_hip_nobar.so): how much of the kernel time is the lock-step of the
# 16 waves between FFT passes - the ceiling of any scheme that replaces workgroup barriers by wave-scope synchronisation
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/ab_nobar
SO=sound-spaces_amd/csrc/libss_hip.so
cp $SO /tmp/new.so
for V in tree nobar tree nobar; do
  cp /tmp/new.so $SO; [ $V = nobar ] && cp gpurun_in/libss_hip_nobar.so $SO
  for C in headline cfg2; do
  A=""; [ $C = cfg2 ] && A="--config cfg2"
  timeout 600 python bench.py $A --no-cpu-baseline --no-plugin-path --no-secondary > gpurun_out/ab_nobar/bench_${V}_$C.json 2>/dev/null
  python - $V $C <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/ab_nobar/bench_{sys.argv[1]}_{sys.argv[2]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], 'overlap value',round(d['value']/1e6,3),'ms',d['ms_per_step'], 'preplanned', round(d['preplanned_single_stream']['value']/1e6,3), 'kernel ms', d['roofline']['avg_launch_ms'])
PY
  done
done
cp /tmp/new.so $SO

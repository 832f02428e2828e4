#!/bin/bash
# build the reference library first (here, no GPU needed): check the previous commit out into a scratch worktree and
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I<worktree>/include <worktree>/sound-spaces_amd/csrc/ss_hip.hip -o gpurun_in/libss_hip_old.so
# (gpurun_in/ is git-ignored and travels to the GPU box with the snapshot)
# same-box A/B of two builds of libss_hip.so: gpurun_in/libss_hip_old.so (built from the previous commit) vs the in-tree one
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/ab_so
SO=sound-spaces_amd/csrc/libss_hip.so
cp $SO /tmp/new.so
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/ab_so/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/ab_so/pytest.log
for V in new old new old; do
  cp /tmp/new.so $SO; [ $V = old ] && cp gpurun_in/libss_hip_old.so $SO
  for C in headline cfg2; do
  A=""; [ $C = cfg2 ] && A="--config cfg2"
  timeout 600 python bench.py $A --no-cpu-baseline --no-plugin-path > gpurun_out/ab_so/bench_${V}_$C.json 2>/dev/null
  python - $V $C <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/ab_so/bench_{sys.argv[1]}_{sys.argv[2]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], 'value',round(d['value']/1e6,3),'ms',d['ms_per_step'], 'ctx_single',round(d['ctx_single_stream']['value']/1e6,3), 'preplanned', round(d['preplanned_single_stream']['value']/1e6,3), d['roofline']['avg_launch_ms'], 'spectral', round(d['spectral_bank']['value']/1e6,3))
PY
  done
done
cp /tmp/new.so $SO

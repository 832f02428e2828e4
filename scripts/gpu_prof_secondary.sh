#!/bin/bash
# kernel traces of the secondary paths: 44.1 kHz bench (general conv kernel + stand-alone spectrogram) and a 2048-unit
# AudioGoal-only / spectrogram-only run (persistent row kernel, k_spectrogram), plus the extension kernels
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sec" -o b44k -- python $GRAFT_REPO_ROOT/bench.py --sr 44100 --no-cpu-baseline --steps 50 --bank-mib 768 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sec" -o k2048 -- python $GRAFT_REPO_ROOT/scripts/kbench.py --sizes 2048 --reps 50 > /dev/null 2>&1
cat > /tmp/ext.py <<'PY'
import sys, os, torch
root = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [root, os.path.join(root, "sound-spaces_amd")]
from ss_amd import ops, planning as P
x = torch.randn((2048, 2, 16000), device="cuda:0")
s, w, _ = P.mel_filterbank_sparse(16000, 64)
ms, mw = torch.from_numpy(s).cuda(), torch.from_numpy(w).cuda()
for _ in range(20):
    ops.logmel(x, ms, mw); ops.gccphat(x); ops.intensity(x)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sec" -o ext -- python /tmp/ext.py > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
for f in $(find gpurun_out/prof_sec -name "*kernel_stats.csv"); do echo "## $f"; grep "ssk::" $f | cut -d, -f1-8; done

#!/bin/bash
# extensions: parity tests + timings of log-mel and GCC-PHAT next to the pooled spectrogram
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 300 python - <<'PY'
import sys, os, numpy as np, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "sound-spaces_amd")]
from ss_amd import ops, planning as P
dev = "cuda:0"
for sr, N in ((16000, 128), (16000, 2048), (44100, 512)):
    x = torch.randn((N, 2, sr), device=dev)
    s, w, _ = P.mel_filterbank_sparse(sr, 64)
    ms, mw = torch.from_numpy(s).to(dev), torch.from_numpy(w).to(dev)
    out = ops.logmel(x, ms, mw); sg = ops.spectrogram(x); gc = ops.gccphat(x)
    for name, fn in (("logmel", lambda: ops.logmel_into(x, out, ms, mw)), ("gccphat", lambda: ops.gccphat_into(x, gc)),
                     ("spectrogram", lambda: ops.spectrogram_into(x, sg))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"sr={sr} N={N} {name}: {us:.1f} us  ({x.numel() * 4 / us / 1e3:.0f} GB/s of input)")
PY

#!/usr/bin/env python
"""k_features<logmel,gccphat> alone, 60 launches on 256 units (for rocprofv3 passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import torch
from ss_amd import ops, planning as P
dev, N, sr = "cuda:0", 256, 16000
x = torch.randn((N, 2, sr), device=dev) * 0.1
ms, mw, _ = P.mel_filterbank_sparse(sr, 64)
ms, mw = torch.from_numpy(ms).to(dev), torch.from_numpy(mw).to(dev)
T = 1 + sr // 160
lm = torch.empty((N, 64, T, 2), device=dev); gc = torch.empty((N, 65, T), device=dev)
for _ in range(60):
    ops.audio_features_into(x, None, lm, gc, ms, mw)
torch.cuda.synchronize()

#!/usr/bin/env python
"""k_features against the three stand-alone feature kernels on one waveform batch (HIP events): us per launch.
usage: python scripts/kbench_features.py [--units 256] [--sr 16000] [--reps 100]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from ss_amd import ops, planning as P
ap = argparse.ArgumentParser()
ap.add_argument("--units", type=int, default=256)
ap.add_argument("--sr", type=int, default=16000)
ap.add_argument("--reps", type=int, default=100)
a = ap.parse_args()
dev = "cuda:0"
N, sr = a.units, a.sr
x = torch.randn((N, 2, sr), device=dev) * 0.1
ms, mw, _ = P.mel_filterbank_sparse(sr, 64)
ms, mw = torch.from_numpy(ms).to(dev), torch.from_numpy(mw).to(dev)
T = 1 + sr // 160
sg = torch.empty((N,) + P.spectrogram_shape(sr), device=dev); lm = torch.empty((N, 64, T, 2), device=dev); gc = torch.empty((N, 65, T), device=dev)


def timeit(fn):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / a.reps * 1e3, 2)


res = {"units": N, "sr": sr}
for rd in range(2):
    res.setdefault("k_spectrogram", []).append(timeit(lambda: ops.spectrogram_into(x, sg)))
    res.setdefault("k_logmel", []).append(timeit(lambda: ops.logmel_into(x, lm, ms, mw)))
    res.setdefault("k_gccphat", []).append(timeit(lambda: ops.gccphat_into(x, gc)))
    res.setdefault("k_features<logmel,gccphat>", []).append(timeit(lambda: ops.audio_features_into(x, None, lm, gc, ms, mw)))
    res.setdefault("k_features<logmel>", []).append(timeit(lambda: ops.audio_features_into(x, None, lm, None, ms, mw)))
    res.setdefault("k_features<gccphat>", []).append(timeit(lambda: ops.audio_features_into(x, None, None, gc)))
    res.setdefault("k_features<spectrogram,logmel,gccphat>", []).append(timeit(lambda: ops.audio_features_into(x, sg, lm, gc, ms, mw)))
    res.setdefault("k_features<spectrogram>", []).append(timeit(lambda: ops.audio_features_into(x, sg)))
print(json.dumps(res))

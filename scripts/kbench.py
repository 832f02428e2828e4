#!/usr/bin/env python
"""Per-kernel timings (HIP events on the launch stream) for the three kernels at a few batch sizes.
usage: python scripts/kbench.py [--sr 16000] [--reps 50] [--sizes 128,2048] [--only fused|conv|spec]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from bench import synth_rir_bank_device
from oracle import ss_oracle as O
from ss_amd import ops
from ss_amd.renderer import BatchedAudioRenderer, RirBank

ap = argparse.ArgumentParser()
ap.add_argument("--sr", type=int, default=16000)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--sizes", default="128,2048")
ap.add_argument("--only", default="")
ap.add_argument("--bank-mib", type=int, default=512)
ap.add_argument("--sounds", type=int, default=102, help="source clips (102 = bench.py: 13 MB of window spectra, more than one XCD's L2)")
a = ap.parse_args()
dev = torch.device("cuda:0")
sr = a.sr
rng = np.random.default_rng(0)
r = BatchedAudioRenderer(sr, device=dev)
for i, c in enumerate(O.synth_sources(rng, sr, k=a.sounds)):
    r.add_source(str(i), c)
R = max(8, (a.bank_mib << 20) // (2 * sr * 4))
r.set_rir_bank(RirBank(synth_rir_bank_device(torch, R, sr, sr, dev, 3), torch.full((R,), sr, dtype=torch.int32, device=dev)))

def timeit(fn, reps):
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for N in [int(x) for x in a.sizes.split(",")]:
    descs = [r.plan_arrays(rng.integers(0, a.sounds, N), np.zeros(N, np.int64), rng.integers(0, R, N)) for _ in range(8)]
    ag = torch.empty((N, 2, sr), device=dev); sg = torch.empty((N,) + r.spectrogram_shape, device=dev)
    res = {}
    if a.only in ("", "fused"): res["fused"] = timeit(lambda k: r.render(descs[k % 8], spectrogram_out=sg, audiogoal_out=(ag if sr > 16384 else None)), a.reps)
    if a.only in ("", "conv"): res["conv"] = timeit(lambda k: r.render_audiogoal(descs[k % 8], out=ag), a.reps)
    if a.only in ("", "spec"): res["spec"] = timeit(lambda k: ops.spectrogram_into(ag, sg), a.reps)
    print(f"N={N} sr={sr} " + " ".join(f"{k}={v:.1f}us" for k, v in res.items()), flush=True)

#!/usr/bin/env python
"""Throughput of the fused launch when consecutive vector steps alternate between S HIP streams (the next step's
workgroups fill the CUs that the previous step's tail and RIR-load phase leave idle).  usage: kbench_streams.py [--n 128]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from bench import synth_rir_bank_device
from oracle import ss_oracle as O
from ss_amd.renderer import BatchedAudioRenderer, RirBank

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=128)
ap.add_argument("--reps", type=int, default=400)
a = ap.parse_args()
dev = torch.device("cuda:0"); sr = 16000
rng = np.random.default_rng(0)
r = BatchedAudioRenderer(sr, device=dev)
for i, c in enumerate(O.synth_sources(rng, sr, k=16)):
    r.add_source(str(i), c)
R = (512 << 20) // (2 * sr * 4)
r.set_rir_bank(RirBank(synth_rir_bank_device(torch, R, sr, sr, dev, 3), torch.full((R,), sr, dtype=torch.int32, device=dev)))
N = a.n
descs = [r.plan_arrays(rng.integers(0, 16, N), np.zeros(N, np.int64), rng.integers(0, R, N)) for _ in range(8)]
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    sg = [torch.empty((N,) + r.spectrogram_shape, device=dev) for _ in range(S)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.reps):
            with torch.cuda.stream(streams[k % S]):
                r.render(descs[k % 8], spectrogram_out=sg[k % S])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps * 1e6
    print(f"N={N} streams={S}: {dt:.1f} us/step  {N / dt:.2f} M env-steps/s", flush=True)

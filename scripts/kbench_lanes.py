#!/usr/bin/env python
"""Do two fused launches on two HIP streams really run side by side?  Raw ctypes calls (3 us of host per launch) of
ss_audio_obs_f32 alternating between S streams, per step size; with an -DSS_AB library SS_HIP_PARTS_LOG2 forces the number of
workgroups per row (0 = one).  Reads: us per launch at S = 1 (the kernel) and at S = 2, 3 (how much of it overlaps).
usage: kbench_lanes.py [--sizes 16,32,64] [--reps 400]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from bench import synth_rir_bank_device
from oracle import ss_oracle as O
from ss_amd import _lib
from ss_amd.renderer import BatchedAudioRenderer, RirBank

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="16,32,64")
ap.add_argument("--reps", type=int, default=400)
ap.add_argument("--sr", type=int, default=16000)
a = ap.parse_args()
dev = torch.device("cuda:0"); sr = a.sr
rng = np.random.default_rng(0)
r = BatchedAudioRenderer(sr, device=dev)
for i, c in enumerate(O.synth_sources(rng, sr, k=102)):
    r.add_source(str(i), c)
R = (256 << 20) // (2 * sr * 4)
r.set_rir_bank(RirBank(synth_rir_bank_device(torch, R, sr, sr, dev, 3), torch.full((R,), sr, dtype=torch.int32, device=dev)))
LIB = _lib.load()
cap = r.rirs.cap
for N in [int(x) for x in a.sizes.split(",")]:
    descs = [r.plan_arrays(rng.integers(0, 102, N), np.zeros(N, np.int64), rng.integers(0, R, N)) for _ in range(8)]
    out = {}
    for S in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(S)]
        sgs = [torch.empty((N,) + r.spectrogram_shape, device=dev) for _ in range(S)]
        calls = []
        for k in range(8 * S):
            d, st, sg = descs[k % 8], streams[k % S], sgs[k % S]
            args = (r._spec.data_ptr(), r.rirs.data.data_ptr(), r.rirs.lengths.data_ptr(), d.desc.data_ptr(), None, sg.data_ptr(),
                    N, 2 * cap, cap, 1, cap, r.n_valid, r.out_len, 0, d.flags, st.cuda_stream)
            calls.append(args)
        fn = LIB.ss_audio_obs_f32
        for _ in range(200):
            for c in calls: fn(*c)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for k in range(a.reps):
                fn(*calls[k % len(calls)])
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / a.reps * 1e6)
        out[S] = best
    print(f"N={N} sr={sr} parts={os.environ.get('SS_HIP_PARTS_LOG2', 'auto')} " + " ".join(f"S={S}:{v:.1f}us" for S, v in out.items()), flush=True)

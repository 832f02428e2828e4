#!/usr/bin/env python
"""BASELINE configs[4]-shaped run: 256 envs, 21 sounds of 1-20 s (multi-second windowing branches), a distractor on every
unit (two convolutions + add, fused general kernel), audiogoal AND spectrogram written, then the log-mel and GCC-PHAT
extension kernels on the audiogoal.  Per-kernel timings with HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from bench import synth_rir_bank_device
from oracle import ss_oracle as O
from ss_amd import ops, planning as P
from ss_amd.renderer import BatchedAudioRenderer, RirBank, UnitRequest

dev = torch.device("cuda:0"); sr, N = 16000, 256
rng = np.random.default_rng(0)
r = BatchedAudioRenderer(sr, device=dev)
secs = rng.integers(1, 21, 21)
for i, s in enumerate(secs):
    r.add_source(str(i), O.synth_sources(rng, sr, k=1, seconds=int(s))[0])
R = (512 << 20) // (2 * sr * 4)
r.set_rir_bank(RirBank(synth_rir_bank_device(torch, R, sr, sr, dev, 3), torch.full((R,), sr, dtype=torch.int32, device=dev)))
plans = []
for _ in range(6):
    units = []
    for n in range(N):
        s_ = int(rng.integers(0, 21)); idx = int(rng.integers(0, secs[s_]))
        units.append(UnitRequest(s_, P.window_start_sim(int(secs[s_]) * sr, sr, idx), int(rng.integers(0, R)),
                                 dis_sound=int(rng.integers(0, 21)), dis_rir=int(rng.integers(0, R))))
    plans.append(r.plan(units))
ag = torch.empty((N, 2, sr), device=dev); sg = torch.empty((N,) + r.spectrogram_shape, device=dev)
s, w, _ = P.mel_filterbank_sparse(sr, 64)
ms, mw = torch.from_numpy(s).to(dev), torch.from_numpy(w).to(dev)
lm = ops.logmel(ag, ms, mw); gc = ops.gccphat(ag)

def timeit(fn, reps=60):
    for k in range(5): fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

t_obs = timeit(lambda k: r.render(plans[k % 6], spectrogram_out=sg, audiogoal_out=ag))
t_lm = timeit(lambda k: ops.logmel_into(ag, lm, ms, mw))
t_gc = timeit(lambda k: ops.gccphat_into(ag, gc))
tot = t_obs + t_lm + t_gc
print(f"cfg5: 256 envs, distractor on: audiogoal+spectrogram {t_obs:.1f} us, log-mel {t_lm:.1f} us, GCC-PHAT {t_gc:.1f} us "
      f"-> {N / tot:.2f} M env-steps/s for all four outputs ({N / t_obs:.2f} M/s for the reference's two)")

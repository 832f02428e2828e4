#!/usr/bin/env python
"""Soak test of the context's overlap mode under window-cache churn: the same random steps (multi-second clips -> many
(sound, t0) keys, a cache of a few dozen entries -> evictions every step, distractors on some steps) go through contexts with two,
four and three overlap lanes, eight steps in flight between joins, and through a plain single-stream context; every step's outputs must be
BIT-IDENTICAL (same kernels, same inputs - anything else is a race: a window spectrum overwritten while a step in flight
still reads it, a descriptor slot reused too early).  usage: soak_ctx.py [rounds] [units]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from oracle import ss_oracle as O
from ss_amd.context import AudioContext
from ss_amd.renderer import RirBank
from ss_amd import planning as P

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sr, dev, K = 16000, "cuda:0", 8
rng = np.random.default_rng(11)
secs = [1, 1, 3, 5, 8, 2, 1, 20, 20, 20, 17, 19, 20, 13]          # 167 (sound, second) keys against a cache of <= 64 entries
src = [O.synth_sources(rng, sr, k=1, seconds=s_)[0] for s_ in secs]
R = 256
bank = RirBank(torch.from_numpy(O.synth_rir(rng, sr, n=R)).to(dev), torch.full((R,), sr, dtype=torch.int32, device=dev))
ctxs = []
LANES = (2, 4, 3, 1)                                               # the last one is the single-stream reference
for lanes in LANES:
    c = AudioContext(sr, max_window_sets=64)
    for i, s_ in enumerate(src):
        c.add_source(f"s{i}", s_)
    c.set_rir_bank(bank.data, bank.lengths)
    c.set_overlap(lanes)
    ctxs.append(c)
shape = ctxs[0].spectrogram_shape
sg = [[torch.empty((N,) + shape, device=dev) for _ in range(K)] for _ in ctxs]
ag = [[torch.empty((N, 2, sr), device=dev) for _ in range(K)] for _ in ctxs]
t_start = time.perf_counter()
bad = 0
for rd in range(rounds):
    steps = []
    for k in range(K):
        snd = rng.integers(0, len(src), N)
        idx = np.array([rng.integers(0, secs[s_]) for s_ in snd])
        t0 = np.array([P.window_start_sim(len(src[s_]), sr, int(i_)) for s_, i_ in zip(snd, idx)])
        rir = rng.integers(0, R, N)
        rir[rng.random(N) < 0.05] = -1
        kw = {}
        if (rd + k) % 3 == 0:
            kw = dict(dis_sound=rng.integers(0, len(src), N), dis_rir=np.where(rng.random(N) < 0.5, rng.integers(0, R, N), -1))
        steps.append((snd, t0, rir, kw))
    for ci, c in enumerate(ctxs):
        for k, (snd, t0, rir, kw) in enumerate(steps):
            want_ag = (k % 2 == 0)
            c.observe(snd, t0, rir, spectrogram_out=sg[ci][k], audiogoal_out=ag[ci][k] if want_ag else None, **kw)
        c.join()
    torch.cuda.synchronize()
    for ci in range(len(ctxs) - 1):
        for k in range(K):
            if not torch.equal(sg[ci][k], sg[-1][k]) or (k % 2 == 0 and not torch.equal(ag[ci][k], ag[-1][k])):
                bad += 1
                print(f"round {rd} step {k}: {LANES[ci]}-lane and single-stream outputs differ "
                      f"(max |d sg| {float((sg[ci][k] - sg[-1][k]).abs().max()):.3e})", flush=True)
    if bad > 5:
        break
st = ctxs[0].stats()
print(f"soak: {rounds} rounds x {K} steps x {N} units, {bad} mismatching steps, {time.perf_counter() - t_start:.1f} s; "
      f"window cache of the overlapped context: {st}")
sys.exit(1 if bad else 0)

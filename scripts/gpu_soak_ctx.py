#!/usr/bin/env python
"""Soak of the context API's step pipeline (descriptor ring, in-place / uploaded descriptors, window cache) on the GPU:
thousands of random steps queued without host syncs on two streams, each compared afterwards, bit for bit, with the
Python-planned renderer (which the parity tests pin to the oracle).  usage: gpu_soak_ctx.py [seed] [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from oracle import ss_oracle as O
from ss_amd import ops
from ss_amd.context import AudioContext
from ss_amd.renderer import BatchedAudioRenderer, RirBank, UnitRequest

dev = "cuda:0"
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
t_start, n_steps, n_units = time.time(), 0, 0
worst = [0.0]
for rnd in range(rounds):
    rng = np.random.default_rng(1000 * seed + rnd)
    sr = 44100 if rnd % 4 == 3 else 16000                       # every fourth round: the fused-rows kernel (k_obs_rows)
    overlap = rnd % 5 in (1, 2, 4)                              # overlap mode: consecutive steps on 2 / 3 / 4 internal lanes
    t4 = 69 if sr == 44100 else 26
    secs = [1, 1, 1, 2, 4][:int(rng.integers(2, 6))]
    many_keys = rnd % 3 == 1                                    # dozens of (sound, second) windows against a small cache:
    if many_keys:                                               # evictions under the in-flight guard, both lane modes
        secs = [int(x) for x in rng.integers(1, 7, 14)]
    src = [O.synth_sources(rng, sr, k=1, seconds=s)[0] for s in secs]
    n_rir = int(rng.integers(4, 40))
    long_rir = rnd % 3 == 2
    rirs = [np.ascontiguousarray(O.synth_rir(rng, sr, length=int(rng.integers(300, (2 * sr if long_rir else sr) + 1)), n=1)[0].T)
            for _ in range(n_rir)]
    bank = RirBank.from_arrays(rirs, dev)
    r = BatchedAudioRenderer(sr, device=dev)
    ctx = AudioContext(sr, max_window_sets=8 if many_keys else int(rng.choice([4, 8, 64])))
    for i, s in enumerate(src):
        r.add_source(f"s{i}", s)
        ctx.add_source(f"s{i}", s)
    r.set_rir_bank(bank)
    ctx.set_rir_bank(bank.data, bank.lengths)
    ctx.set_overlap((2, 3, 4)[rnd % 3] if overlap else 1)     # (the caller changes streams under it: busy streams take the fence)
    spectral = rnd % 2 == 1
    if spectral:
        spectra = ops.rir_spectra(bank.data)
        ctx.set_rir_spectra(spectra)
        r.rirs.spectra = spectra
    distract = bool(rng.integers(0, 2))
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    which = 0
    for chunk in range(8):
        steps, outs = [], []
        torch.cuda.synchronize()
        for k in range(40):
            n = int(rng.choice([1, 2, 3, 5] if many_keys else [1, 2, 5, 17, 64, 128, 129, 256, 257, 400] if sr == 16000
                               else [1, 2, 5, 17, 64, 128, 130]))
            sound = rng.integers(0, len(src), n)
            t0 = np.array([int(rng.integers(0, secs[s])) * sr if secs[s] > 1 else 0 for s in sound])
            rir = rng.integers(-1, n_rir, n)
            cols = dict(sound=sound, t0=t0, rir=rir)
            if distract:
                cols.update(dis_sound=rng.integers(0, min(3, len(src)), n), dis_rir=np.where(rng.random(n) < 0.7, rng.integers(0, n_rir, n), -1))
            if rng.random() < 0.3:
                which ^= 1
            want_ag = rng.random() < 0.4
            sg = torch.empty((n, 65, t4, 2), device=dev)
            ag = torch.empty((n, 2, sr), device=dev) if want_ag else None
            with torch.cuda.stream(streams[which]):
                ctx.observe(spectrogram_out=sg, audiogoal_out=ag, **cols)
            steps.append(cols); outs.append((sg, ag))
        if overlap:
            ctx.join()
        torch.cuda.synchronize()
        for cols, (sg, ag) in zip(steps, outs):
            n = len(cols["sound"])
            if distract:
                units = [UnitRequest(int(cols["sound"][i]), int(cols["t0"][i]), int(cols["rir"][i]), silent=cols["rir"][i] < 0,
                                     dis_sound=int(cols["dis_sound"][i]) if cols["dis_rir"][i] >= 0 else -1,
                                     dis_rir=int(cols["dis_rir"][i])) for i in range(n)]
                plan = r.plan(units)
            else:
                plan = r.plan_arrays(cols["sound"], cols["t0"], cols["rir"])
            ag2, sg2 = r.render(plan, want_audiogoal=ag is not None)
            def same(a, b, what):
                if not torch.equal(a, b):                       # (a different launch-flag choice may differ in the last bit)
                    err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
                    assert err <= 2e-6, (what, rnd, chunk, n, err)
                    worst[0] = max(worst[0], err)
            if ag is not None:
                same(ag, ag2, "audiogoal")
            same(sg, sg2, "spectrogram")
            n_steps += 1; n_units += n
    print(f"round {rnd}: sr={sr} overlap={overlap} many_keys={many_keys} spectral={spectral} distract={distract} long_rir={long_rir} ok; cache {ctx.stats()}", flush=True)
print(f"SOAK OK: {n_steps} steps, {n_units} units, worst non-identical relerr {worst[0]:.2e}, {time.time() - t_start:.1f} s")

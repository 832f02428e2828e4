#!/usr/bin/env python
"""Randomised parity soak on the GPU: many random batches (sizes 1..600, ragged RIRs, silent units, distractors,
multi-second clips, both kernels) against the oracle.  Prints the worst relative error per category."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from oracle import ss_oracle as O
from ss_amd import planning as P
from ss_amd.renderer import BatchedAudioRenderer, RirBank, UnitRequest

dev = "cuda:0"
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
worst = {"audiogoal": 0.0, "spectrogram": 0.0, "audiogoal_only": 0.0}
t_start = time.time()
n_checked = 0
for rnd in range(rounds):
    rng = np.random.default_rng(seed0 * 1000 + rnd)
    sr = 16000
    n_src, n_rir = int(rng.integers(1, 5)), int(rng.integers(1, 9))
    secs = [int(rng.integers(1, 4)) for _ in range(n_src)]
    src = [O.synth_sources(rng, sr, k=1, seconds=s)[0] for s in secs]
    rirs = []
    for _ in range(n_rir):
        L = int(rng.choice([sr, int(rng.integers(200, sr)), int(rng.integers(sr, 2 * sr))]))
        rirs.append(np.ascontiguousarray(O.synth_rir(rng, sr, length=L, n=1)[0].T))
    distract = bool(rng.integers(0, 2))
    N = int(rng.choice([1, 2, 7, 33, 128, 129, 300, 600]))
    r = BatchedAudioRenderer(sr, device=dev)
    for i, s in enumerate(src):
        r.add_source(f"s{i}", s)
    r.set_rir_bank(RirBank.from_arrays(rirs, dev))
    units, meta = [], []
    for n in range(N):
        s_, h_ = int(rng.integers(0, n_src)), int(rng.integers(0, n_rir))
        idx = int(rng.integers(0, secs[s_]))
        if rng.random() < 0.05:
            units.append(UnitRequest(s_, 0, h_, silent=True)); meta.append(None); continue
        t0 = P.window_start_sim(len(src[s_]), sr, idx)
        if distract and rng.random() < 0.3:
            ds, dh = int(rng.integers(0, n_src)), int(rng.integers(0, n_rir))
            units.append(UnitRequest(s_, t0, h_, dis_sound=ds, dis_rir=dh)); meta.append((s_, h_, idx, ds, dh))
        else:
            units.append(UnitRequest(s_, t0, h_)); meta.append((s_, h_, idx, None, None))
    plan = r.plan(units)
    ag, sg = r.render(plan, want_audiogoal=True)
    ag2 = r.render_audiogoal(plan)
    ag, sg, ag2 = ag.cpu().numpy(), sg.cpu().numpy(), ag2.cpu().numpy()
    cache = {}
    pick = range(N) if N <= 40 else sorted(set(rng.integers(0, N, 40).tolist()) | {0, N - 1})
    for n in pick:
        m = meta[n]
        if m is None:
            assert not ag[n].any() and not sg[n].any() and not ag2[n].any()
            continue
        if m not in cache:
            s_, h_, idx, ds, dh = m
            if ds is None:
                a = O.compute_audiogoal(src[s_], rirs[h_], sr, audio_index=idx)
            else:
                dsrc = src[ds][:sr] if len(src[ds]) >= sr else src[ds]
                a = O.compute_audiogoal(src[s_], rirs[h_], sr, audio_index=idx, distractor=src[ds], distractor_rir=rirs[dh])
            cache[m] = (a, O.compute_spectrogram(a.astype(np.float32)))
        a, s = cache[m]
        worst["audiogoal"] = max(worst["audiogoal"], O.relerr(ag[n], a))
        worst["audiogoal_only"] = max(worst["audiogoal_only"], O.relerr(ag2[n], a))
        worst["spectrogram"] = max(worst["spectrogram"], O.relerr(sg[n], s))
        n_checked += 1
    del r
print(f"soak seed {seed0}: {rounds} batches, {n_checked} units checked in {time.time() - t_start:.0f} s; worst rel err {worst}")
assert max(worst.values()) <= 1e-4, worst


# ---- 44.1 kHz (partitioned rows, unfused spectrogram) and SoundSpaces-2.0 stepping (0.25-s steps, wrapping index) ----
worst2 = {"44k_audiogoal": 0.0, "44k_spectrogram": 0.0, "ss2_audiogoal": 0.0, "ss2_spectrogram": 0.0}
for rnd in range(max(2, rounds // 6)):
    rng = np.random.default_rng(seed0 * 7919 + rnd)
    sr = 44100
    src = [O.synth_sources(rng, sr, k=1, seconds=int(rng.integers(1, 3)))[0] for _ in range(2)]
    rirs = [np.ascontiguousarray(O.synth_rir(rng, sr, length=int(rng.integers(1000, sr + 1)), n=1)[0].T) for _ in range(3)]
    r = BatchedAudioRenderer(sr, device=dev)
    for i, s_ in enumerate(src):
        r.add_source(f"s{i}", s_)
    r.set_rir_bank(RirBank.from_arrays(rirs, dev))
    N = int(rng.choice([1, 5, 40, 140]))
    units, meta = [], []
    for n in range(N):
        s_, h_ = int(rng.integers(0, 2)), int(rng.integers(0, 3))
        idx = int(rng.integers(0, len(src[s_]) // sr))
        units.append(UnitRequest(s_, P.window_start_sim(len(src[s_]), sr, idx), h_)); meta.append((s_, h_, idx))
    ag, sg = r.render(r.plan(units), want_audiogoal=True)
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    cache = {}
    for n in sorted(set(rng.integers(0, N, min(N, 12)).tolist())):
        m = meta[n]
        if m not in cache:
            a = O.compute_audiogoal(src[m[0]], rirs[m[1]], sr, audio_index=m[2])
            cache[m] = (a, O.compute_spectrogram(a.astype(np.float32)))
        worst2["44k_audiogoal"] = max(worst2["44k_audiogoal"], O.relerr(ag[n], cache[m][0]))
        worst2["44k_spectrogram"] = max(worst2["44k_spectrogram"], O.relerr(sg[n], cache[m][1]))
    del r
    # SS2.0
    sr = 16000
    # 3-s clips (1-s ones are tiled x3 at load, continuous_simulator.py:408-410): the early branch never meets the clip end
    src = [O.tile_short_source(O.synth_sources(rng, sr, k=1, seconds=int(rng.choice([1, 3])))[0], sr) for _ in range(2)]
    rirs = [np.ascontiguousarray(O.synth_rir(rng, sr, length=int(rng.integers(500, 40000)), n=1)[0].T) for _ in range(3)]
    r = BatchedAudioRenderer(sr, device=dev, step_time=0.25, wrap=True)
    for i, s_ in enumerate(src):
        r.add_source(f"s{i}", s_)
    r.set_rir_bank(RirBank.from_arrays(rirs, dev))
    N = int(rng.choice([3, 50, 200]))
    units, meta = [], []
    for n in range(N):
        s_, h_ = int(rng.integers(0, 2)), int(rng.integers(0, 3))
        si = int(rng.integers(0, len(src[s_])))
        units.append(UnitRequest(s_, P.window_start_continuous(si), h_)); meta.append((s_, h_, si))
    ag, sg = r.render(r.plan(units), want_audiogoal=True)
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    for n in sorted(set(rng.integers(0, N, min(N, 12)).tolist())):
        s_, h_, si = meta[n]
        a = O.convolve_with_rir(src[s_], rirs[h_], sr, si, 0.25)
        worst2["ss2_audiogoal"] = max(worst2["ss2_audiogoal"], O.relerr(ag[n], a))
        worst2["ss2_spectrogram"] = max(worst2["ss2_spectrogram"], O.relerr(sg[n], O.compute_spectrogram(a.astype(np.float32))))
    del r
print(f"soak seed {seed0} (44.1 kHz, SS2.0): worst rel err {worst2}")
assert max(worst2.values()) <= 1e-4, worst2

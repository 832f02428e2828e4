"""Randomised parity sweep of the product path against the oracle (test infrastructure: the oracle is the checker here).

Every trial draws a sampling rate, a handful of sources (1-s clips and multi-second clips whose length is not a multiple
of the rate), RIRs of ragged lengths (a few samples ... 2.4 s, i.e. also longer than the observation), a step of 1-400
units with random (sound, audio index, RIR, silent, distractor) and renders it from BOTH bank forms, with and without the
waveform; every unit is compared with `oracle.compute_audiogoal` (simulator.py:608-666) and `compute_spectrogram`
(nav.py:86-100) at the north-star tolerance (1e-4 of the unit's peak; exact zeros for silent units).

    python scripts/gpu_fuzz.py --trials 200 --seed 1 [--out profiles/r6/fuzz.txt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sound-spaces_amd"))

from oracle import ss_oracle as O                      # noqa: E402
from ss_amd import planning as P                       # noqa: E402
from ss_amd.renderer import BatchedAudioRenderer, RirBank, UnitRequest   # noqa: E402

TOL = 1e-4
RATES = [16000, 16000, 16000, 44100, 44100, 48000, 22050, 32000, 8000, 11025]


def draw_trial(rng):
    sr = int(rng.choice(RATES))
    n_src = int(rng.integers(1, 5))
    srcs = []
    for _ in range(n_src):
        kind = rng.integers(0, 3)
        n = sr if kind == 0 else int(rng.integers(2, 6)) * sr if kind == 1 else int(rng.uniform(2.0, 5.5) * sr)
        srcs.append((rng.standard_normal(n) * rng.uniform(0.05, 0.5)).astype(np.float32))
    n_rir = int(rng.integers(1, 7))
    rirs = []
    for _ in range(n_rir):
        kind = rng.integers(0, 5)
        if kind == 0:
            L = int(rng.integers(1, 64))
        elif kind == 1:
            L = int(rng.uniform(1.0, 2.4) * sr)
        else:
            L = int(rng.uniform(0.05, 1.0) * sr)
        if kind == 0:                                          # a few taps (synth_rir's decay model needs a real length)
            rirs.append((rng.standard_normal((L, 2)) * 0.3).astype(np.float32))
            continue
        h = O.synth_rir(rng, sr, length=L, n=1)[0]             # [2, L]
        rirs.append(np.ascontiguousarray(h.T).astype(np.float32))
    n_units = int(rng.choice([1, 2, 3, 5, 7, 10, 16, 31, 32, 33, 42, 43, 64, 96, 97, 128, 150, 257, 400, 700, 1100],
                             p=None))
    if sr >= 32000:
        n_units = min(n_units, 300 if rng.random() < 0.1 else 160)
    with_dis = rng.random() < 0.4
    units, keys = [], []
    for _ in range(n_units):
        s = int(rng.integers(0, n_src))
        idx = 0 if len(srcs[s]) == sr else int(rng.integers(0, len(srcs[s]) // sr))
        h = int(rng.integers(0, n_rir))
        silent = rng.random() < 0.05
        ds = dh = -1
        if with_dis and rng.random() < 0.5:
            ds, dh = int(rng.integers(0, n_src)), int(rng.integers(0, n_rir))
        units.append(UnitRequest(s, P.window_start_sim(len(srcs[s]), sr, idx), h, silent, ds, dh))
        keys.append((s, idx, h, silent, ds, dh))
    return sr, srcs, rirs, units, keys


def draw_continuous(rng):
    """SoundSpaces 2.0 steps (continuous_simulator.py:413-456): a step of `step_time` seconds from a random sample index, the
    early / steady / wrapping branch by the RIR's length, CROSSFADE with the previous step's RIR for some units."""
    sr = int(rng.choice(RATES))
    step_time = float(rng.choice([0.25, 0.25, 0.1, 0.5, 1.0]))
    n_src = int(rng.integers(1, 4))
    srcs = []
    for _ in range(n_src):
        kind = rng.integers(0, 3)
        n = sr if kind == 0 else int(rng.integers(2, 5)) * sr if kind == 1 else int(rng.uniform(2.0, 4.5) * sr)
        srcs.append(O.tile_short_source((rng.standard_normal(n) * rng.uniform(0.05, 0.5)).astype(np.float32), sr))
    n_rir = int(rng.integers(2, 7))
    rirs = []
    for _ in range(n_rir):
        L = int(rng.uniform(1.0, 2.0) * sr) if rng.integers(0, 5) == 0 else int(rng.uniform(0.02, 1.0) * sr)
        rirs.append(np.ascontiguousarray(O.synth_rir(rng, sr, length=L, n=1)[0].T).astype(np.float32))
    n_units = int(rng.choice([1, 2, 3, 5, 10, 16, 33, 42, 64, 97, 128]))
    if sr >= 32000:
        n_units = min(n_units, 64)
    with_fade = rng.random() < 0.5
    units, keys = [], []
    for _ in range(n_units):
        s = int(rng.integers(0, n_src))
        idx = int(rng.integers(0, len(srcs[s])))
        if rng.random() < 0.2:                                  # near the clip's end: the wrapping branch
            idx = len(srcs[s]) - int(rng.integers(1, int(sr * step_time) + 1))
        h = int(rng.integers(0, n_rir))
        silent = rng.random() < 0.05
        last = int(rng.integers(0, n_rir)) if with_fade and rng.random() < 0.6 else -1
        units.append(UnitRequest(s, idx, h, silent, wrap=idx - len(rirs[h]) >= 0, last_rir=last,
                                 last_wrap=(idx - len(rirs[last]) >= 0) if last >= 0 else None))
        keys.append((s, idx, h, silent, last))
    return sr, step_time, srcs, rirs, units, keys


def run_continuous(rng, dev):
    sr, step_time, srcs, rirs, units, keys = draw_continuous(rng)
    refs = {}
    for k in set(keys):
        s, idx, h, silent, last = k
        a = np.asarray(O.compute_audiogoal_continuous(srcs[s], rirs[h], sr, idx, step_time, rirs[last] if last >= 0 else None,
                                                      last >= 0, silent), np.float64)
        refs[k] = (a, O.compute_spectrogram(a.astype(np.float32)))
    worst = 0.0
    for spectral in (False, True):
        r = BatchedAudioRenderer(sr, device=dev, step_time=step_time, wrap=True)
        for i, s in enumerate(srcs):
            r.add_source(f"s{i}", s)
        r.set_rir_bank(RirBank.from_arrays(rirs, dev))
        if spectral:
            r.rirs.build_spectra()
        ag, sg = r.render(r.plan(units), want_audiogoal=True)
        sg2 = r.render(r.plan(units))[1]
        ag, sg, sg2 = ag.cpu().numpy(), sg.cpu().numpy(), sg2.cpu().numpy()
        assert not (np.isnan(ag).any() or np.isnan(sg).any() or np.isnan(sg2).any()), "NaN"
        for n, k in enumerate(keys):
            ra, rs = refs[k]
            if k[3]:
                assert not ag[n].any() and not sg[n].any() and not sg2[n].any(), f"silent unit {n} not zero"
                continue
            for got, ref, what in ((ag[n], ra, "audiogoal"), (sg[n], rs, "spectrogram"), (sg2[n], rs, "spectrogram-only")):
                scale = np.abs(ref).max()
                err = np.abs(got - ref).max() / scale if scale > 0 else np.abs(got).max()
                worst = max(worst, err)
                assert err <= TOL, f"unit {n} {what} spectral={spectral}: {err:.3e} key={k} step_time={step_time} lens=" \
                                   f"{len(srcs[k[0]])},{len(rirs[k[2]])}"
    return sr, len(units), len(srcs), len(rirs), any(k[4] >= 0 for k in keys), worst


def run_engine(rng, dev):
    """Several consecutive steps through AudioEngine (RirStore / BucketedRirStore + the C++ context, `observe_columns`) with a
    store smaller than the pool of poses: entries are evicted and rewritten between steps, rows grow, the spectral rows of
    rewritten entries are rebuilt; every unit of every step against the oracle."""
    from ss_amd.renderer import AudioEngine
    sr = int(rng.choice([16000, 16000, 44100, 22050, 48000]))
    n_src = int(rng.integers(1, 4))
    srcs = []
    for _ in range(n_src):
        n = sr if rng.integers(0, 2) == 0 else int(rng.integers(2, 5)) * sr
        srcs.append((rng.standard_normal(n) * rng.uniform(0.05, 0.5)).astype(np.float32))
    pool = int(rng.integers(4, 40))
    rirs = [np.ascontiguousarray(O.synth_rir(rng, sr, length=int(rng.uniform(0.05, 1.6 if rng.random() < 0.15 else 1.0) * sr),
                                             n=1)[0].T).astype(np.float32) for _ in range(pool)]
    per_step = int(rng.choice([1, 2, 5, 10, 16, 32, 64]))
    if sr >= 32000:
        per_step = min(per_step, 32)
    slots = int(rng.integers(max(2, min(pool, per_step)), pool + 4))
    kw = {}
    form = int(rng.integers(0, 3))
    if form == 1:
        kw["rir_spectral"] = False
    elif form == 2 and rng.random() < 0.5:
        kw["rir_buckets"] = [(slots, sr // 2), (slots, sr), (slots, 2 * sr)]
    with_dis = rng.random() < 0.3
    eng = AudioEngine(sr, device=dev, rir_slots=slots, **kw)
    sids = [eng.source_id(f"s{i}", s) for i, s in enumerate(srcs)]
    worst, n_total = 0.0, 0
    # overlap mode (ss_ctx_set_overlap): the steps alternate between 2-3 internal streams and NOTHING is read back before the last
    # step was issued; entries are evicted and rewritten while earlier steps are still in flight - the engine's stores order their
    # device writes behind the lanes themselves (RirStore.before_device_write); the lanes' fences, the shared window-spectra pool
    # and the descriptor ring are covered as well
    lanes = int(rng.choice([2, 3])) if rng.random() < 0.4 else 1
    if lanes > 1:
        eng.context().set_overlap(lanes)
    pending = []
    for step in range(int(rng.integers(3, 9))):
        n = int(rng.integers(1, per_step + 1))
        picks = rng.choice(pool, size=min(n, slots // (2 if with_dis else 1), pool), replace=False)
        eng.begin_batch()
        cols = dict(sound=[], t0=[], rir=[])
        if with_dis:
            cols.update(dis_sound=[], dis_rir=[])
        keys = []
        for u in range(n):
            s = int(rng.integers(0, n_src))
            idx = 0 if len(srcs[s]) == sr else int(rng.integers(0, len(srcs[s]) // sr))
            h = int(picks[u % len(picks)])
            silent = rng.random() < 0.05
            cols["sound"].append(sids[s]); cols["t0"].append(P.window_start_sim(len(srcs[s]), sr, idx))
            cols["rir"].append(-1 if silent else eng.rir_slot(("pose", h), (lambda h=h: rirs[h])))
            ds = dh = -1
            if with_dis:
                if rng.random() < 0.6 and not silent:
                    ds, dh = int(rng.integers(0, n_src)), int(picks[(u + 1) % len(picks)])
                cols["dis_sound"].append(sids[ds] if ds >= 0 else -1)
                cols["dis_rir"].append(eng.rir_slot(("pose", dh), (lambda dh=dh: rirs[dh])) if ds >= 0 else -1)
            keys.append((s, idx, h, silent, ds, dh))
        cols = {k: np.asarray(v, np.int64) for k, v in cols.items()}
        sg = torch.full((n,) + tuple(O.spectrogram_shape(sr)), float("nan"), device=dev)
        ag = torch.full((n, 2, sr), float("nan"), device=dev)
        eng.observe_columns(cols, spectrogram_out=sg, audiogoal_out=ag if step % 2 == 0 else None)
        pending.append((step, n, keys, sg, ag if step % 2 == 0 else None))
    if lanes > 1:
        eng.context().join()
    torch.cuda.synchronize()
    for step, n, keys, sg, ag in pending:
        sg = sg.cpu().numpy()
        ag = ag.cpu().numpy() if ag is not None else None
        assert not np.isnan(sg).any() and (ag is None or not np.isnan(ag).any()), "NaN / unwritten output rows"
        for u, k in enumerate(keys):
            s, idx, h, silent, ds, dh = k
            if silent:
                assert not sg[u].any() and (ag is None or not ag[u].any()), f"step {step} silent unit {u} not zero"
                continue
            ra = np.asarray(O.compute_audiogoal(srcs[s], rirs[h], sr, idx, False, srcs[ds] if ds >= 0 else None,
                                                rirs[dh] if ds >= 0 else None), np.float64)
            rs = O.compute_spectrogram(ra.astype(np.float32))
            for got, ref, what in ((ag[u] if ag is not None else None, ra, "audiogoal"), (sg[u], rs, "spectrogram")):
                if got is None:
                    continue
                err = np.abs(got - ref).max() / np.abs(ref).max()
                worst = max(worst, err)
                assert err <= TOL, f"step {step} unit {u} {what}: {err:.3e} key={k} store={type(eng.store).__name__} " \
                                   f"slots={slots} pool={pool} spectral={eng.rir_spectral} lanes={lanes}"
        n_total += n
    return sr, n_total, n_src, pool, with_dis, worst


def run_trial(rng, dev):
    sr, srcs, rirs, units, keys = draw_trial(rng)
    pad = "reflect" if rng.random() < 0.7 else "constant"      # librosa < 0.10 (the reference's era) / >= 0.10 centre padding
    refs = {}
    for k in set(keys):
        s, idx, h, silent, ds, dh = k
        a = O.compute_audiogoal(srcs[s], rirs[h], sr, idx, silent, srcs[ds] if ds >= 0 else None,
                                rirs[dh] if ds >= 0 else None)
        a = np.asarray(a, np.float64)
        refs[k] = (a, O.compute_spectrogram(a.astype(np.float32), pad_mode=pad))
    worst = 0.0
    for spectral in (False, True):
        r = BatchedAudioRenderer(sr, device=dev, pad_mode=pad)
        for i, s in enumerate(srcs):
            r.add_source(f"s{i}", s)
        r.set_rir_bank(RirBank.from_arrays(rirs, dev))
        if spectral:
            r.rirs.build_spectra()
        ag, sg = r.render(r.plan(units), want_audiogoal=True)
        sg2 = r.render(r.plan(units))[1]
        ag, sg, sg2 = ag.cpu().numpy(), sg.cpu().numpy(), sg2.cpu().numpy()
        assert not (np.isnan(ag).any() or np.isnan(sg).any() or np.isnan(sg2).any()), "NaN"
        for n, k in enumerate(keys):
            ra, rs = refs[k]
            if k[3]:
                assert not ag[n].any() and not sg[n].any() and not sg2[n].any(), f"silent unit {n} not zero"
                continue
            for got, ref, what in ((ag[n], ra, "audiogoal"), (sg[n], rs, "spectrogram"), (sg2[n], rs, "spectrogram-only")):
                scale = np.abs(ref).max()
                err = np.abs(got - ref).max() / scale if scale > 0 else np.abs(got).max()
                worst = max(worst, err)
                assert err <= TOL, f"unit {n} {what} spectral={spectral}: {err:.3e} key={k} lens=" \
                                   f"{len(srcs[k[0]])},{len(rirs[k[2]])}"
    return sr, len(units), len(srcs), len(rirs), any(k[4] >= 0 for k in keys), worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mode", choices=["sim", "continuous", "engine"], default="sim",
                    help="sim: SoundSpacesSim._compute_audiogoal steps; continuous: SoundSpaces 2.0 steps (distractor column = cross-fade)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = "cuda:0"
    lines, fails, worst_all = [], 0, 0.0
    t_start = time.time()
    for t in range(args.trials):
        rng = np.random.default_rng([args.seed, t])
        try:
            sr, n, ns, nr, dis, worst = {'sim': run_trial, 'continuous': run_continuous, 'engine': run_engine}[args.mode](rng, dev)
            worst_all = max(worst_all, worst)
            lines.append(f"trial {t:4d} ok   sr={sr:5d} units={n:3d} sources={ns} rirs={nr} distractor={int(dis)} worst={worst:.2e}")
        except Exception as e:                          # noqa: BLE001 - a sweep reports every failing trial
            fails += 1
            lines.append(f"trial {t:4d} FAIL {type(e).__name__}: {e}")
        print(lines[-1], flush=True)
    tail = f"# mode {args.mode}, {args.trials} trials, seed {args.seed}: {fails} failed, worst relative error {worst_all:.2e} " \
           f"(tolerance {TOL:.0e}), {time.time() - t_start:.0f} s"
    print(tail)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines + [tail]) + "\n")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()

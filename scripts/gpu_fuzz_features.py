"""Randomised parity sweep of the waveform-side entry points against the oracle (test infrastructure).

Every trial draws N (1-300), a row length (300 ... 70 000 samples: odd lengths, lengths not a multiple of 4 or of the hop, rows
shorter than one STFT frame's reach), a pad mode, a sampling rate / mel-band count for the filter bank and GCC-PHAT's lag range,
fills [N, 2, len] with noise of random level (some rows silent, some ears silent) and compares `ops.spectrogram`
(nav.py:86-100), `ops.logmel`, `ops.gccphat`, every subset of `ops.audio_features` (k_features) and `ops.intensity`
(avwan_sensors.py:91-100) with the oracle, every row, at 1e-4 of the row's peak (GCC-PHAT: of its full scale 1.0).

    python scripts/gpu_fuzz_features.py --trials 200 --seed 1 [--out profiles/r6/fuzz_features.txt]
"""
import argparse
import itertools
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sound-spaces_amd"))

from oracle import ss_oracle as O                      # noqa: E402
from ss_amd import ops, planning as P                  # noqa: E402

TOL = 1e-4
DEV = "cuda:0"


def rel(got, ref, scale=None):
    scale = np.abs(ref).max() if scale is None else scale
    return float(np.abs(got - ref).max() / scale) if scale > 0 else float(np.abs(got).max())


def run_trial(rng):
    N = int(rng.choice([1, 2, 3, 5, 16, 31, 64, 128, 129, 256, 300]))
    kind = rng.integers(0, 4)
    n = (int(rng.integers(300, 3000)) if kind == 0 else int(rng.integers(3000, 70000)) if kind == 1
         else int(rng.choice([16000, 44100, 48000, 22050, 4000, 16001, 15999])))
    if N * n > 8_000_000:
        N = max(1, 8_000_000 // n)
    pad = str(rng.choice(["reflect", "constant"]))
    sr = int(rng.choice([16000, 44100, 48000, 22050]))
    n_mels = int(rng.choice([64, 40, 32]))                  # (bands wider than 64 bins - 20-band banks - are outside the ABI's
                                                            #  stated limit, include/ss_hip.h: SS_EINVAL)
    max_lag = int(rng.choice([32, 16, 8, 1]))
    x = (rng.standard_normal((N, 2, n)) * rng.uniform(1e-3, 1.0, (N, 1, 1))).astype(np.float32)
    if N > 2:
        x[int(rng.integers(0, N))] = 0.0                     # a silent row
        x[int(rng.integers(0, N)), int(rng.integers(0, 2))] = 0.0      # a silent ear
    xd = torch.from_numpy(x).to(DEV)
    ms, mw, _ = P.mel_filterbank_sparse(sr, n_mels)
    msd, mwd = torch.from_numpy(ms).to(DEV), torch.from_numpy(mw).to(DEV)
    got = {"spectrogram": ops.spectrogram(xd, pad).cpu().numpy(), "logmel": ops.logmel(xd, msd, mwd, 1e-6, pad).cpu().numpy(),
           "gccphat": ops.gccphat(xd, max_lag, 1e-8, pad).cpu().numpy()}
    fused = {}
    names = ("spectrogram", "logmel", "gccphat")
    for k in range(1, 4):
        for want in itertools.combinations(names, k):
            out = ops.audio_features(xd, want, msd, mwd, 1e-6, max_lag, 1e-8, pad)
            fused[want] = {w: out[w].cpu().numpy() for w in want}
    inten = ops.intensity(xd).cpu().numpy() if n >= 150 else None
    worst = 0.0
    rows = range(N) if N <= 48 else sorted(set(rng.integers(0, N, 48).tolist()))
    for i in rows:
        ref = {"spectrogram": O.compute_spectrogram(x[i], pad_mode=pad), "logmel": O.compute_logmel(x[i], sr, n_mels, 1e-6, pad),
               "gccphat": O.compute_gcc_phat(x[i], max_lag, 1e-8, pad)}
        # PHAT divides every bin by its own magnitude: a bin that is empty to 1e-4 of the frame's median (white noise through
        # a reflect-padded, hence symmetric, first frame has them) carries an arbitrary unit phase in ANY float32 evaluation -
        # frames with such a bin are left out of the comparison (1 frame in ~10^4 here)
        G = np.abs(O.stft(x[i, 0], pad_mode=pad) * np.conj(O.stft(x[i, 1], pad_mode=pad)))
        well = G.min(axis=0) > 1e-4 * np.median(G, axis=0)
        for w in names:
            scale = 1.0 if w == "gccphat" else None
            if w == "gccphat" and not (x[i, 0].any() and x[i, 1].any()):
                continue                                       # 0 / (0 + eps): both sides are exact zeros or eps-noise
            if w == "gccphat":
                ref[w] = ref[w][:, well]
                got_w = got[w][i][:, well]
            else:
                got_w = got[w][i]
            e = rel(got_w, ref[w], scale)
            assert e <= TOL, f"{w} row {i}: {e:.3e} (N={N} n={n} pad={pad} sr={sr} mels={n_mels} lag={max_lag})"
            worst = max(worst, e)
            for want, outs in fused.items():
                if w in outs:
                    e = rel(outs[w][i][:, well] if w == "gccphat" else outs[w][i], ref[w], scale)
                    assert e <= TOL, f"audio_features{want}.{w} row {i}: {e:.3e} (N={N} n={n} pad={pad} sr={sr} mels={n_mels} lag={max_lag})"
                    worst = max(worst, e)
        if inten is not None and x[i].max() > 0:
            r = float(O.intensity(x[i])[0])
            e = abs(float(inten[i]) - r) / max(r, 1e-30)
            assert e <= TOL, f"intensity row {i}: {e:.3e} (N={N} n={n})"
    return N, n, pad, sr, n_mels, max_lag, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    ops.init()
    lines, fails, worst_all = [], 0, 0.0
    t_start = time.time()
    for t in range(args.trials):
        rng = np.random.default_rng([args.seed, t])
        try:
            N, n, pad, sr, n_mels, max_lag, worst = run_trial(rng)
            worst_all = max(worst_all, worst)
            lines.append(f"trial {t:4d} ok   N={N:3d} len={n:5d} pad={pad:8s} sr={sr} mels={n_mels} lag={max_lag:2d} worst={worst:.2e}")
        except Exception as e:                          # noqa: BLE001 - a sweep reports every failing trial
            fails += 1
            lines.append(f"trial {t:4d} FAIL {type(e).__name__}: {e}")
        print(lines[-1], flush=True)
    tail = f"# features, {args.trials} trials, seed {args.seed}: {fails} failed, worst relative error {worst_all:.2e} " \
           f"(tolerance {TOL:.0e}), {time.time() - t_start:.0f} s"
    print(tail)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines + [tail]) + "\n")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()

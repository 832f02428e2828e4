"""Debug helper for scripts/gpu_fuzz_features.py: re-run ONE trial's GCC-PHAT and print where the error sits."""
import sys, os, warnings
warnings.simplefilter("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
import gpu_fuzz_features as m
from oracle import ss_oracle as O
from ss_amd import ops

seed, trial = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([seed, trial])
N = int(rng.choice([1, 2, 3, 5, 16, 31, 64, 128, 129, 256, 300]))
kind = rng.integers(0, 4)
n = (int(rng.integers(300, 3000)) if kind == 0 else int(rng.integers(3000, 70000)) if kind == 1
     else int(rng.choice([16000, 44100, 48000, 22050, 4000, 16001, 15999])))
if N * n > 8_000_000:
    N = max(1, 8_000_000 // n)
pad = str(rng.choice(["reflect", "constant"]))
sr = int(rng.choice([16000, 44100, 48000, 22050]))
n_mels = int(rng.choice([64, 40, 32, 20]))
max_lag = int(rng.choice([32, 16, 8, 1]))
x = (rng.standard_normal((N, 2, n)) * rng.uniform(1e-3, 1.0, (N, 1, 1))).astype(np.float32)
print("N", N, "n", n, pad, "lag", max_lag)
xd = torch.from_numpy(x).to("cuda:0")
for lag in (max_lag, 32):
    g = ops.gccphat(xd, lag, 1e-8, pad).cpu().numpy()
    f = ops.audio_features(xd, ("gccphat",), None, None, 1e-6, lag, 1e-8, pad)["gccphat"].cpu().numpy()
    for i in range(N):
        ref = O.compute_gcc_phat(x[i], lag, 1e-8, pad)
        e, ef = np.abs(g[i] - ref), np.abs(f[i] - ref)
        if e.max() > 3e-5 or ef.max() > 3e-5:
            j = np.unravel_index(e.argmax(), e.shape)
            print(f"lag {lag} row {i}: level {np.abs(x[i]).max():.2e} standalone {e.max():.2e} at {j} fused {ef.max():.2e}; frame-wise max", np.sort(e.max(axis=0))[-4:])
            # the frame's spectra: the smallest |X_l||X_r| bin
            fr = j[1]
            Xl = O.stft(x[i, 0], pad_mode=pad)[:, fr]; Xr = O.stft(x[i, 1], pad_mode=pad)[:, fr]
            G = np.abs(Xl * np.conj(Xr))
            print("    smallest |G| / median |G|:", np.sort(G)[:3] / np.median(G), "|Xl| min/med", np.abs(Xl).min() / np.median(np.abs(Xl)), "|Xr| min/med", np.abs(Xr).min() / np.median(np.abs(Xr)))

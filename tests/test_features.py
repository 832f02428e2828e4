"""k_features / ss_audio_features_f32: every STFT-derived feature of BASELINE.json configs[4] ("GCC-PHAT + log-mel fused
sensor"; extensions) and the reference's pooled spectrogram (soundspaces/tasks/nav.py:86-100) from ONE pass
over the binaural waveform.  Checkers: the oracle (compute_spectrogram restates the reference; compute_logmel /
compute_gcc_phat are the textbook definitions, "parity unpinned" by nature) and the three stand-alone kernels."""
import numpy as np
import pytest

from oracle import ss_oracle as O
from ss_amd import planning as P

TOL = 1e-4      # max|got - ref| <= TOL * max|ref|, on EVERY unit


def check(got, ref, tol=TOL):
    assert got.shape == ref.shape and not np.isnan(got).any()
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()


def binaural(rng, n_units, n, quiet_head=True):
    """correlated ears (a delayed, scaled copy + noise): GCC-PHAT has a peak to find; unit 1 starts silent"""
    base = rng.standard_normal((n_units, n + 40)).astype(np.float32)
    x = np.stack([base[:, 20:20 + n], 0.7 * base[:, 13:13 + n] + 0.05 * rng.standard_normal((n_units, n))], axis=1)
    x = (x * rng.uniform(1e-2, 1.0, (n_units, 1, 1))).astype(np.float32)
    if quiet_head and n_units > 1:
        x[1, :, : n // 3] = 0.0
    return x


@pytest.mark.parametrize("n,sr,n_mels,gpw", [(16000, 16000, 64, 1), (16000, 16000, 40, 7), (4000, 16000, 64, 2),
                                             (44100, 44100, 64, 5), (15999, 16000, 64, 3), (9000, 48000, 32, 2)])
def test_hostsim_features_vs_oracle(n, sr, n_mels, gpw):
    from hostsim import hs
    rng = np.random.default_rng(n + n_mels)
    x = binaural(rng, 2, n)
    for pm, name in ((0, "reflect"), (1, "constant")):
        got = hs.features(x, sr, n_mels=n_mels, pad_mode=pm, gpw=gpw)
        for k in range(2):
            check(got["spectrogram"][k], O.compute_spectrogram(x[k], pad_mode=name))
            check(got["logmel"][k], O.compute_logmel(x[k], sr, n_mels=n_mels, pad_mode=name))
            check(got["gccphat"][k], O.compute_gcc_phat(x[k], pad_mode=name))


def test_hostsim_features_subsets_equal_the_full_launch_and_the_stand_alone_kernels():
    from hostsim import hs
    rng = np.random.default_rng(5)
    x = binaural(rng, 2, 16000)
    full = hs.features(x, 16000)
    for want in (("logmel",), ("gccphat",), ("spectrogram",), ("logmel", "gccphat")):
        part = hs.features(x, 16000, want=want)
        assert set(part) == set(want)
        for k in want:
            np.testing.assert_array_equal(part[k], full[k])
    np.testing.assert_allclose(full["spectrogram"], hs.spectrogram(x), rtol=0, atol=2e-6 * np.abs(full["spectrogram"]).max())
    np.testing.assert_allclose(full["logmel"], hs.logmel(x, 16000), rtol=0, atol=2e-5)
    np.testing.assert_allclose(full["gccphat"], hs.gccphat(x), rtol=0, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n_units,n_mels", [(16000, 256, 64), (44100, 24, 64), (16000, 37, 40)])
def test_gpu_features_vs_oracle_and_stand_alone_kernels(sr, n_units, n_mels):
    import torch
    from ss_amd import ops
    dev = "cuda:0"
    rng = np.random.default_rng(sr + n_units)
    x = binaural(rng, n_units, sr)
    x[0] = 0.0                                                       # a silent unit
    start, w, _ = P.mel_filterbank_sparse(sr, n_mels)
    ms, mw = torch.from_numpy(start).to(dev), torch.from_numpy(w).to(dev)
    xd = torch.from_numpy(x).to(dev)
    got = ops.audio_features(xd, want=("spectrogram", "logmel", "gccphat"), mel_start=ms, mel_w=mw)
    sg, lm, gc = (got[k].cpu().numpy() for k in ("spectrogram", "logmel", "gccphat"))
    # the stand-alone kernels on the same waveform (each re-reads it and redoes the STFT)
    sg1, lm1, gc1 = ops.spectrogram(xd).cpu().numpy(), ops.logmel(xd, ms, mw).cpu().numpy(), ops.gccphat(xd).cpu().numpy()
    assert np.abs(sg - sg1).max() <= 2e-6 * np.abs(sg1).max()
    assert np.abs(lm - lm1).max() <= 5e-5 and np.abs(gc - gc1).max() <= 5e-6
    assert not sg[0].any() and np.allclose(lm[0], np.log(1e-6), rtol=1e-6)
    for k in range(n_units):                                         # EVERY unit against the oracle, 1e-4 of its peak
        if k == 0:
            continue
        check(sg[k], O.compute_spectrogram(x[k]))
        check(lm[k], O.compute_logmel(x[k], sr, n_mels=n_mels))
        check(gc[k], O.compute_gcc_phat(x[k]))
    only = ops.audio_features(xd, want=("gccphat",))
    assert set(only) == {"gccphat"} and torch.equal(only["gccphat"], got["gccphat"])
    with pytest.raises(Exception):                                   # a bank wider than the fused kernel serves: SS_EINVAL
        s2, w2, _ = P.mel_filterbank_sparse(sr, 128)
        ops.audio_features(xd, want=("logmel",), mel_start=torch.from_numpy(s2).to(dev), mel_w=torch.from_numpy(w2).to(dev))

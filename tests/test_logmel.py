"""EXTENSION: log-mel front end, ss_logmel_f32.  The reference has no such sensor, so the checker
is the oracle's textbook definition (oracle/ss_oracle.py: compute_logmel, "parity unpinned"); the band-sparse filter bank
the product builds (ss_amd.planning) is checked against the oracle's dense matrix, the kernel against the oracle."""
import numpy as np
import pytest

from oracle import ss_oracle as O
from ss_amd import planning as P

# log() is ill-conditioned near eps: compare in the domain the tolerance is defined on (relative to the largest value)
TOL = 1e-4


def check(got, ref, tol=TOL):
    assert got.shape == ref.shape and not np.isnan(got).any()
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()


@pytest.mark.parametrize("sr,n_mels", [(16000, 64), (44100, 64), (16000, 40), (16000, 128), (44100, 128)])
def test_sparse_filterbank_equals_dense_definition(sr, n_mels):
    start, w, max_len = P.mel_filterbank_sparse(sr, n_mels)
    dense = np.zeros((n_mels, 257))
    for j in range(n_mels):
        n = min(max_len, 257 - start[j])
        dense[j, start[j]:start[j] + n] = w[j, :n]
        assert not w[j, n:].any()
    ref = O.mel_filterbank(sr, n_mels)
    np.testing.assert_allclose(dense, ref, rtol=1e-6, atol=1e-9)
    assert max_len <= 60 and max_len % 4 == 0 and n_mels * max_len <= 4096
    assert (start >= 0).all() and (start <= 256).all() and not (start % 4).any()


def test_mel_scale_round_trip_and_known_points():
    f = np.array([0.0, 200.0, 1000.0, 4000.0, 8000.0])
    np.testing.assert_allclose(O.mel_to_hz(O.hz_to_mel(f)), f, rtol=1e-12, atol=1e-9)
    assert abs(float(O.hz_to_mel(1000.0)) - 15.0) < 1e-12           # Slaney: 15 mel at 1 kHz
    np.testing.assert_allclose(P._slaney_mel(f), O.hz_to_mel(f), rtol=1e-12)


@pytest.mark.parametrize("n,sr,n_mels,gpw", [(16000, 16000, 64, 1), (16000, 16000, 64, 7), (4000, 16000, 40, 2),
                                             (44100, 44100, 64, 5), (15999, 16000, 128, 3), (9000, 48000, 32, 2)])
def test_hostsim_kernel_vs_oracle(n, sr, n_mels, gpw):
    from hostsim import hs
    rng = np.random.default_rng(n + n_mels)
    x = (rng.standard_normal((2, 2, n)) * np.array([1.0, 0.01])[:, None, None]).astype(np.float32)
    x[1, :, : n // 3] = 0.0                                         # silent head: log(eps) rows
    for pm, name in ((0, "reflect"), (1, "constant")):
        got = hs.logmel(x, sr, n_mels=n_mels, pad_mode=pm, gpw=gpw)
        assert got.shape == (2, n_mels, 1 + n // 160, 2)
        for k in range(2):
            check(got[k], O.compute_logmel(x[k], sr, n_mels=n_mels, pad_mode=name))


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n_units,n_mels", [(16000, 37, 64), (44100, 9, 64), (16000, 300, 40)])
def test_gpu_kernel_vs_oracle(sr, n_units, n_mels):
    import torch
    from ss_amd import ops
    dev = "cuda:0"
    rng = np.random.default_rng(sr + n_units)
    x = rng.standard_normal((n_units, 2, sr)).astype(np.float32) * rng.uniform(1e-3, 1.0, (n_units, 1, 1)).astype(np.float32)
    x[0] = 0.0
    start, w, _ = P.mel_filterbank_sparse(sr, n_mels)
    ms, mw = torch.from_numpy(start).to(dev), torch.from_numpy(w).to(dev)
    xd = torch.from_numpy(x).to(dev)
    got = ops.logmel(xd, ms, mw).cpu().numpy()
    got2 = torch.ops.ss_hip.logmel(xd, ms, mw, 1e-6, 0).cpu().numpy()
    np.testing.assert_array_equal(got, got2)
    assert np.allclose(got[0], np.log(1e-6), rtol=1e-6)             # silent unit
    for k in list(range(min(n_units, 6))) + [n_units - 1]:
        check(got[k], O.compute_logmel(x[k], sr, n_mels=n_mels))
    with pytest.raises(Exception):
        ops.logmel(xd, ms, torch.zeros((n_mels, 68), device=dev))    # max_len > 64 -> SS_EINVAL

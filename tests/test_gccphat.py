"""EXTENSION: GCC-PHAT between the ears, ss_gccphat_f32.  Not in the reference; the checker is the
oracle's textbook definition (oracle/ss_oracle.py: compute_gcc_phat, "parity unpinned") plus domain properties: a pure
inter-aural delay puts the peak at that lag, and swapping the ears mirrors the lag axis."""
import numpy as np
import pytest

from oracle import ss_oracle as O

TOL = 1e-4


def check(got, ref, tol=TOL):
    assert got.shape == ref.shape and not np.isnan(got).any()
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()


def delayed_pair(rng, n, d):
    s = rng.standard_normal(n + 64).astype(np.float32)
    return np.stack([s[32 - d:32 - d + n], s[32:32 + n]])        # left = right delayed by d samples


def test_oracle_properties():
    rng = np.random.default_rng(0)
    for d in (0, 3, -7, 20):
        g = O.compute_gcc_phat(delayed_pair(rng, 16000, d), max_lag=32)
        assert g.shape == (65, 101)
        assert (np.argmax(g[:, 5:-5], axis=0) == 32 + d).all()
    x = rng.standard_normal((2, 8000)).astype(np.float32)
    np.testing.assert_allclose(O.compute_gcc_phat(x[::-1], 16), O.compute_gcc_phat(x, 16)[::-1], atol=1e-12)


@pytest.mark.parametrize("n,max_lag,gpw", [(16000, 32, 1), (16000, 8, 7), (4000, 32, 2), (44100, 20, 5), (15999, 1, 3)])
def test_hostsim_kernel_vs_oracle(n, max_lag, gpw):
    from hostsim import hs
    rng = np.random.default_rng(n + max_lag)
    x = rng.standard_normal((2, 2, n)).astype(np.float32)
    x[1] = delayed_pair(rng, n, 5) * 0.05
    for pm, name in ((0, "reflect"), (1, "constant")):
        got = hs.gccphat(x, max_lag=max_lag, pad_mode=pm, gpw=gpw)
        for k in range(2):
            check(got[k], O.compute_gcc_phat(x[k], max_lag=max_lag, pad_mode=name))
    assert not hs.gccphat(np.zeros((1, 2, 4000), np.float32)).any()          # silence -> exact zeros


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n_units,max_lag", [(16000, 37, 32), (44100, 9, 16), (16000, 300, 4)])
def test_gpu_kernel_vs_oracle(sr, n_units, max_lag):
    import torch
    from ss_amd import ops
    dev = "cuda:0"
    rng = np.random.default_rng(sr + n_units)
    x = rng.standard_normal((n_units, 2, sr)).astype(np.float32)
    x[1] = delayed_pair(rng, sr, -3)
    x[2] = 0.0
    xd = torch.from_numpy(x).to(dev)
    got = ops.gccphat(xd, max_lag).cpu().numpy()
    np.testing.assert_array_equal(got, torch.ops.ss_hip.gccphat(xd, max_lag, 1e-8, 0).cpu().numpy())
    assert not got[2].any()
    assert (np.argmax(got[1][:, 5:-5], axis=0) == max_lag - 3).all()
    for k in list(range(min(n_units, 6))) + [n_units - 1]:
        check(got[k], O.compute_gcc_phat(x[k], max_lag=max_lag))
    with pytest.raises(Exception):
        ops.gccphat(xd, 33)

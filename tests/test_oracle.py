"""Oracle (oracle/ss_oracle.py) vs vectors produced by the reference's own code
(tests/golden/make_golden.py), plus independent cross-checks of the librosa /
skimage restatements.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ss_oracle as O
from golden_util import golden, case_inputs, case_outputs

CASES = golden()[1]


def oracle_audiogoal(name):
    d = case_inputs(name)
    sr = d["sr"]
    if name == "silent":
        return O.compute_audiogoal(None, None, sr, silent=True)
    if name in ("zero_rir", "empty_rir"):
        src = O.synth_sources(np.random.default_rng(1), sr, k=3)[0]
        return O.compute_audiogoal(src, O.zero_rir(sr), sr)
    if name.startswith("cont_"):
        src3 = O.tile_short_source(d["source"], sr)
        # habitat_sim hands the RIR over as python lists -> np.array(...) is float64
        # (continuous_simulator.py:419), so the SS2.0 reference convolves in float64.
        last = d["last_rir"].astype(np.float64) if "last_rir" in d else None
        return O.compute_audiogoal_continuous(src3, d["rir"].astype(np.float64), sr, d["sample_index"],
                                              d["step_time"], last_rir=last, use_crossfade=last is not None)
    if name.startswith("savi_"):
        return O.compute_audiogoal_savi_dataset(d["source"], d["rir"], sr, d["audio_index"])
    return O.compute_audiogoal(d["source"], d["rir"], sr, audio_index=d.get("audio_index", 0),
                               distractor=d.get("distractor"), distractor_rir=d.get("distractor_rir"))


@pytest.mark.parametrize("name", CASES)
def test_audiogoal_matches_reference_run(name):
    ref, _, stride = case_outputs(name)
    got = oracle_audiogoal(name)
    assert got.shape[0] == 2
    assert got[:, ::stride].shape == ref.shape
    assert got.dtype == ref.dtype          # float64 zeros when silent, float32 otherwise
    np.testing.assert_array_equal(got[:, ::stride], ref)   # same scipy, same calls -> bit-exact


@pytest.mark.parametrize("name", CASES)
def test_spectrogram_matches_reference_composition(name):
    ref_a, ref_s, stride = case_outputs(name)
    got = O.compute_spectrogram(oracle_audiogoal(name))
    assert got.shape == ref_s.shape == O.spectrogram_shape(case_inputs(name)["sr"])
    np.testing.assert_allclose(got, ref_s, rtol=0, atol=1e-6)
    if name in ("silent", "zero_rir", "empty_rir"):
        assert not got.any()               # belief_predictor.py keys "silent" on exact zeros


def test_audio_index_advance():
    _, _, params = golden()
    for name, p in params.items():
        if "next_index" in p:
            assert O.next_audio_index(p["audio_index"], p["seconds"] * p["sr"], p["sr"]) == p["next_index"]
    assert O.next_audio_index(0, 16000, 16000) == 0


def test_shape_known_answers():
    z = golden()[0]
    assert tuple(z["ones16k/spectrogram_shape"]) == (65, 26, 2) == O.spectrogram_shape(16000)
    assert tuple(z["ones44k/spectrogram_shape"]) == (65, 69, 2) == O.spectrogram_shape(44100)
    assert O.compute_spectrogram(np.ones((2, 16000))).shape == (65, 26, 2)


def test_intensity_matches_reference_run():
    z = golden()[0]
    got = O.intensity(oracle_audiogoal("clip1s"))
    np.testing.assert_array_equal(np.asarray(got), z["clip1s/intensity"])


# ---- the unified window formula the HIP kernels implement --------------------

@pytest.mark.parametrize("name", [c for c in CASES if c.startswith(("clip1s", "multi_", "savi_"))
                                  and not c.endswith("44k")])
def test_unified_window_formula(name):
    d = case_inputs(name)
    sr = d["sr"]
    variant = "savi" if name.startswith("savi_") else "sim"
    t0 = O.window_start(d["source"].shape[0], d["rir"].shape[0], sr, d.get("audio_index", 0), variant)
    ref = oracle_audiogoal(name)
    # direct O(L*T) evaluation on three 96-sample output slices
    for a in (0, 7777, sr - 96):
        got = O.conv_window_direct(d["source"], d["rir"], t0 + a, 96)
        assert np.abs(got - ref[:, a:a + 96]).max() < 2e-6 * np.abs(ref).max()


@pytest.mark.parametrize("name", ["cont_early", "cont_steady", "cont_wrap", "cont_early_past_end"])
def test_unified_window_formula_continuous(name):
    d = case_inputs(name)
    sr = d["sr"]
    src3 = O.tile_short_source(d["source"], sr)
    ref = oracle_audiogoal(name)
    ns = int(sr * d["step_time"])
    assert not ref[:, ns:].any()
    # the reference wraps the clip around only in its steady branch (index >= rir length, :438-447); the early branch
    # slices source[:index+num_sample], i.e. reads zeros past the clip end (:433-437)
    wrap = d["sample_index"] - d["rir"].shape[0] >= 0
    for a in (0, 1500, ns - 96):
        got = O.conv_window_direct(src3, d["rir"], d["sample_index"] + a, 96, wrap=wrap)
        assert np.abs(got - ref[:, a:a + 96]).max() < 2e-6 * np.abs(ref).max()
    if name == "cont_early_past_end":           # and wrapping there WOULD be wrong (the case is discriminating)
        bad = O.conv_window_direct(src3, d["rir"], d["sample_index"] + ns - 96, 96, wrap=True)
        assert np.abs(bad - ref[:, ns - 96:ns]).max() > 1e-3 * np.abs(ref).max()


# ---- librosa.stft / skimage.block_reduce restatements, independent checks ----

@pytest.mark.parametrize("sr", [16000, 44100])
@pytest.mark.parametrize("pad_mode", ["reflect", "constant"])
def test_stft_vs_torch(sr, pad_mode):
    rng = np.random.default_rng(11)
    x = rng.standard_normal(sr).astype(np.float32)
    got = O.stft(x, pad_mode=pad_mode)
    assert got.dtype == np.complex64 and got.shape == (257, 1 + sr // 160)
    ref = torch.stft(torch.from_numpy(x).double(), n_fft=512, hop_length=160, win_length=400,
                     window=torch.hann_window(400, periodic=True, dtype=torch.float64),
                     center=True, pad_mode=pad_mode, return_complex=True).numpy()
    assert O.relerr(got, ref) < 5e-7


def test_stft_vs_scipy_shorttimefft():
    from scipy.signal import ShortTimeFFT
    rng = np.random.default_rng(12)
    sr = 16000
    x = rng.standard_normal(sr)
    win = O.stft_window()
    sft = ShortTimeFFT(win, hop=160, fs=sr, fft_mode="onesided", phase_shift=None)
    # zero padding ('constant') with the frame centred on k*hop: p-range [0, 101)
    ref = sft.stft(x, p0=0, p1=101, padding="zeros")
    got = O.stft(x, pad_mode="constant")
    assert ref.shape == got.shape
    assert O.relerr(got, ref) < 1e-12


def test_block_reduce_semantics():
    a = np.arange(257 * 101, dtype=np.float64).reshape(257, 101)
    r = O.block_reduce_mean(a)
    assert r.shape == (65, 26)
    assert r[0, 0] == a[:4, :4].mean()
    assert r[64, 3] == a[256, 12:16].sum() / 16.0        # Nyquist row: 1 real + 3 pad rows
    assert r[5, 25] == a[20:24, 100].sum() / 16.0        # last column: 1 real + 3 pad frames
    assert r[64, 25] == a[256, 100] / 16.0


def test_float32_vs_float64_headroom():
    """scipy's f32 path sits ~1e-7 of max-abs from f64: ample headroom under 1e-4."""
    d = case_inputs("clip1s")
    a32 = O.compute_audiogoal(d["source"], d["rir"], 16000)
    a64 = O.compute_audiogoal(d["source"].astype(np.float64), d["rir"].astype(np.float64), 16000)
    assert O.relerr(a32, a64) < 2e-6

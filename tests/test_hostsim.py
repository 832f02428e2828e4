"""The HIP kernel sources, compiled for the host by tests/hostsim (fibers instead of lanes), against the
oracle and the reference-generated golden vectors.  This checks the index algebra of the kernels (LDS
layouts, radix passes, Hermitian item pairing, partition planning) on CPU; the same cases run on the real
MI355X in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from oracle import ss_oracle as O
from golden_util import golden, case_inputs, case_outputs
from ss_amd import planning as P

hs = pytest.importorskip("hostsim.hs")

TOL = 1e-4      # north-star tolerance: max|got-ref| / max|ref|  (fp32)


def planar(rir_wav, cap=None):
    """[L,2] wav layout -> [1,2,cap] zero-padded planar bank"""
    L = rir_wav.shape[0]
    cap = cap or (L + (L & 1))
    b = np.zeros((1, 2, cap), np.float32)
    b[0, :, :L] = rir_wav.T
    return b


def check(got, ref, tol=TOL):
    assert not np.isnan(got).any()
    assert O.relerr(got, ref) <= tol, O.relerr(got, ref)
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * np.abs(ref).max())


SIM_CASES = [c for c in golden()[1] if c.startswith(("clip1s", "multi_")) and not c.endswith("44k")]


@pytest.mark.parametrize("name", SIM_CASES)
def test_sim_branches_vs_reference_vectors(name):
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    t0 = P.window_start_sim(len(d["source"]), sr, d.get("audio_index", 0))
    out, sg = hs.run([d["source"]], planar(d["rir"]), [d["rir"].shape[0]], [dict(sound=0, t0=t0, rir=0)],
                     sr, sr, fuse=True)
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)


def test_unfused_equals_fused_and_interleaved_layout():
    d = case_inputs("clip1s_ragged")
    sr = d["sr"]
    bank = planar(d["rir"])
    u = [dict(sound=0, t0=0, rir=0)]
    a1, s1 = hs.run([d["source"]], bank, [d["rir"].shape[0]], u, sr, sr, fuse=True)
    a2, s2 = hs.run([d["source"]], bank, [d["rir"].shape[0]], u, sr, sr, fuse=False, want_spectrogram=True)
    a3, _ = hs.run([d["source"]], bank, [d["rir"].shape[0]], u, sr, sr, interleaved=True)
    a4, s4 = hs.run([d["source"]], bank, [d["rir"].shape[0]], u, sr, sr, fuse=True, simple=False)   # loop kernel
    np.testing.assert_array_equal(a1, a2)
    np.testing.assert_array_equal(a1, a3)
    np.testing.assert_array_equal(a1, a4)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(s1, s4)


@pytest.mark.parametrize("wgs", [1, 3, 64])
def test_persistent_row_kernel_equals_one_workgroup_per_row(wgs, fuse=False):
    # k_conv_rows: `wgs` workgroups walk the rows (prefetching the next row's RIR); silent units and empty RIRs in the
    # middle of a walk, ragged RIR lengths, a 0.25-s step (n_valid < out_len)
    rng = np.random.default_rng(7)
    sr = 16000
    srcs = O.synth_sources(rng, sr, k=2)
    bank = np.zeros((4, 2, sr), np.float32)
    lens = [sr, 5000, 0, 12345]
    for i, L in enumerate(lens):
        if L:
            bank[i, :, :L] = O.synth_rir(rng, sr, length=L, n=1)[0]
    units = [dict(sound=0, t0=0, rir=0), dict(rir=-1), dict(sound=1, t0=0, rir=1), dict(sound=0, t0=0, rir=2),
             dict(sound=1, t0=0, rir=3)]
    for n_valid in (sr, 4000):
        a_ref, s_ref = hs.run(srcs, bank, lens, units, n_valid, sr, fuse=fuse, want_spectrogram=True)
        a, sg = hs.run(srcs, bank, lens, units, n_valid, sr, fuse=fuse, want_spectrogram=True, persist=wgs)
        np.testing.assert_array_equal(a, a_ref)
        np.testing.assert_array_equal(sg, s_ref)
        assert not a[1].any() and not a[3].any() and a[4].any()


def test_distractor_silent_and_zero_rir_in_one_batch():
    d = case_inputs("distractor")
    sr = d["sr"]
    bank = np.concatenate([planar(d["rir"]), planar(d["distractor_rir"]), np.zeros((1, 2, sr), np.float32)])
    units = [dict(sound=0, t0=0, rir=0, dis_sound=1, dis_t0=0, dis_rir=1),   # source + distractor
             dict(rir=-1),                                                     # silent (simulator.py:610)
             dict(sound=0, t0=0, rir=2),                                       # unreadable RIR -> zero RIR
             dict(sound=0, t0=0, rir=0)]                                       # plain
    # entry 2 has rir_len 0 ("empty RIR file") and a zero row
    out, sg = hs.run([d["source"], d["distractor"]], bank, [sr, sr, 0], units, sr, sr, fuse=True)
    ref_a, ref_s, stride = case_outputs("distractor")
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)
    for n in (1, 2):
        assert not out[n].any() and not sg[n].any()          # exact zeros (belief_predictor.py keys on them)
    ref_plain, _, st = case_outputs("clip1s")
    check(out[3][:, ::st], ref_plain)


def test_zero_rir_with_full_length():
    d = case_inputs("clip1s")
    sr = d["sr"]
    out, sg = hs.run([d["source"]], np.zeros((1, 2, sr), np.float32), [sr], [dict(sound=0, t0=0, rir=0)],
                     sr, sr, fuse=True)
    assert not out.any() and not sg.any()


@pytest.mark.parametrize("name", ["savi_i0", "savi_i2"])
def test_savi_dataset_variant(name):
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    t0 = P.window_start_savi_dataset(d["rir"].shape[0], sr, d["audio_index"])
    out, sg = hs.run([d["source"]], planar(d["rir"]), [d["rir"].shape[0]], [dict(sound=0, t0=t0, rir=0)],
                     sr, sr, fuse=True)
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)


@pytest.mark.parametrize("name", ["cont_early", "cont_steady", "cont_wrap", "cont_early_past_end"])
def test_continuous_simulator_windows(name):
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    src3 = O.tile_short_source(d["source"], sr)
    ns = int(sr * d["step_time"])
    wrap = d["sample_index"] - d["rir"].shape[0] >= 0         # the reference wraps only in its steady branch
    out, sg = hs.run([src3], planar(d["rir"]), [d["rir"].shape[0]],
                     [dict(sound=0, t0=P.window_start_continuous(d["sample_index"]), rir=0, wrap=wrap)],
                     ns, sr, fuse=True)
    assert not out[0][:, ns:].any()
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)


@pytest.mark.parametrize("fuse", [True, False])
def test_continuous_crossfade_in_kernel(fuse):
    """SS_FLAG_CROSSFADE: previous RIR = term 1, blended in registers by the loop kernel (one launch) -- against the
    vector produced by running the reference's _compute_audiogoal with CROSSFADE on; a unit without a previous RIR
    (first step of an episode) in the same batch stays unblended."""
    d = case_inputs("cont_crossfade")
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs("cont_crossfade")
    src3 = O.tile_short_source(d["source"], sr)
    ns = int(sr * d["step_time"])
    bank = np.concatenate([planar(d["rir"]), planar(d["last_rir"])])
    L = [d["rir"].shape[0], d["last_rir"].shape[0]]
    t0 = P.window_start_continuous(d["sample_index"])
    units = [dict(sound=0, t0=t0, rir=0, wrap=True, last_rir=1), dict(sound=0, t0=t0, rir=0, wrap=True)]
    out, sg = hs.run([src3], bank, L, units, ns, sr, fuse=fuse, want_spectrogram=True, crossfade=True)
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)
    plain = O.convolve_with_rir(src3, d["rir"], sr, d["sample_index"], d["step_time"])
    check(out[1], plain)
    n = int(0.05 * sr)
    assert np.abs(out[0][:, n + 1:] - out[1][:, n + 1:]).max() == 0.0          # beyond the ramp: the current RIR alone
    assert np.abs(out[0][:, :n] - out[1][:, :n]).max() > 0.0


def test_continuous_crossfade_branches_differ_between_the_two_rirs():
    """cont_crossfade_mixed: current RIR (9000 taps) is in the steady branch and wraps around the clip, the previous
    RIR (50000 taps, 4 partition blocks) is in the early branch and reads zeros past the clip end."""
    d = case_inputs("cont_crossfade_mixed")
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs("cont_crossfade_mixed")
    src3 = O.tile_short_source(d["source"], sr)
    ns = int(sr * d["step_time"])
    cap = 50000
    bank = np.concatenate([planar(d["rir"], cap), planar(d["last_rir"], cap)])
    units = [dict(sound=0, t0=d["sample_index"], rir=0, wrap=True, last_rir=1, last_wrap=False)]
    out, sg = hs.run([src3], bank, [d["rir"].shape[0], d["last_rir"].shape[0]], units, ns, sr, fuse=True,
                     crossfade=True)
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)


@pytest.mark.parametrize("name", ["clip1s", "clip1s_ragged", "multi_L1.0_i2", "multi_L1.5_i2", "distractor", "clip1s_44k"])
def test_spectral_rir_bank_equals_time_domain_bank(name):
    """k_conv_spec (RIR block spectra precomputed by the bank builder, no forward FFT per step) against the
    reference-run vectors and against k_conv on the same inputs: loop-free and loop kernel, fused and unfused,
    a 1.5-s RIR (2 blocks), a distractor term, 44.1 kHz (3 x 3 blocks)."""
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    t0 = P.window_start_sim(len(d["source"]), sr, d.get("audio_index", 0))
    srcs, banks, lens = [d["source"]], [d["rir"]], [d["rir"].shape[0]]
    u = dict(sound=0, t0=t0, rir=0)
    if "distractor" in d:
        srcs.append(d["distractor"]); banks.append(d["distractor_rir"]); lens.append(d["distractor_rir"].shape[0])
        u.update(dis_sound=1, dis_t0=0, dis_rir=1)
    cap = max(lens) + (max(lens) & 1)
    bank = np.concatenate([planar(b, cap) for b in banks])
    fuse = sr <= P.KB
    a_s, s_s = hs.run(srcs, bank, lens, [u, dict(rir=-1)], sr, sr, fuse=fuse, want_spectrogram=True, spectral=True)
    a_t, s_t = hs.run(srcs, bank, lens, [u, dict(rir=-1)], sr, sr, fuse=fuse, want_spectrogram=True)
    check(a_s[0][:, ::stride], ref_a)
    check(s_s[0], ref_s)
    assert not a_s[1].any() and not s_s[1].any()                                # silent unit: exact zeros
    assert np.abs(a_s - a_t).max() <= 2e-6 * np.abs(a_t).max()
    if fuse:                                                                     # and the unfused spectral kernel
        a_u, _ = hs.run(srcs, bank, lens, [u], sr, sr, fuse=False, spectral=True)
        assert np.abs(a_u[0] - a_s[0]).max() <= 2e-6 * np.abs(a_t).max()


@pytest.mark.parametrize("wgs", [1, 2, 64])
def test_persistent_spectral_row_kernel_equals_one_workgroup_per_row(wgs):
    """k_conv_spec_rows: `wgs` workgroups walk the units (both ears back to back, next row's H' prefetched); silent units
    and empty RIRs in the middle of a walk, ragged RIRs, a 0.25-s step."""
    rng = np.random.default_rng(17)
    sr = 16000
    srcs = O.synth_sources(rng, sr, k=2)
    bank = np.zeros((4, 2, sr), np.float32)
    lens = [sr, 5000, 0, 12345]
    for i, L in enumerate(lens):
        if L:
            bank[i, :, :L] = O.synth_rir(rng, sr, length=L, n=1)[0]
    units = [dict(sound=0, t0=0, rir=0), dict(rir=-1), dict(sound=1, t0=0, rir=1), dict(sound=0, t0=0, rir=2),
             dict(sound=1, t0=0, rir=3), dict(sound=0, t0=0, rir=1), dict(rir=-1)]
    for n_valid in (sr, 4000):
        a_ref, _ = hs.run(srcs, bank, lens, units, n_valid, sr, spectral=True)
        a, _ = hs.run(srcs, bank, lens, units, n_valid, sr, spectral=True, persist=wgs)
        np.testing.assert_array_equal(a, a_ref)
        assert not a[1].any() and not a[3].any() and a[4].any()


def test_44k_three_output_blocks_three_rir_blocks():
    d = case_inputs("clip1s_44k")
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs("clip1s_44k")
    out, sg = hs.run([d["source"]], planar(d["rir"]), [sr], [dict(sound=0, t0=0, rir=0)], sr, sr,
                     fuse=False, want_spectrogram=True)
    assert sg.shape[1:] == (65, 69, 2)
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)


def test_long_rir_two_blocks_multisecond_steady():
    """RIR longer than one partition block (1.5 s at 16 kHz) in the steady branch: negative offsets m."""
    d = case_inputs("multi_L1.5_i4")
    sr = d["sr"]
    assert d["rir"].shape[0] > P.KB
    ws = P.plan_window_set(len(d["source"]), 4 * sr, 2, 1)
    assert (ws.m_min, ws.count) == (-1, 2)


def test_four_second_rir_four_partition_blocks():
    """SS2.0 ray-traced RIRs run up to 4 s (irTime): 64000 taps = 4 partition blocks at 16 kHz, reaching back over
    four source windows; SS2.0 steady branch with wrap-around and the SS1.0 multi-second steady branch."""
    rng = np.random.default_rng(11)
    sr, L = 16000, 64000
    src = O.synth_sources(rng, sr, k=1, seconds=5)[0]
    h = O.synth_rir(rng, sr, length=L, n=1)[0]
    h *= np.exp(-np.arange(L) / 30000.0)[None, :].astype(np.float32)            # keep the tail audible but decaying
    bank = h[None].astype(np.float32)
    rir_wav = np.ascontiguousarray(h.T)
    # SS2.0: 0.25-s step at sample 70000 of the (wrapping) clip
    ns = 4000
    out, _ = hs.run([src], bank, [L], [dict(sound=0, t0=P.window_start_continuous(70000), rir=0, wrap=True)], ns, sr,
                    simple=False)
    ref = O.convolve_with_rir(src, rir_wav, sr, 70000, 0.25)
    check(out[0], ref)
    # SS1.0 multi-second, audio_index = 4 (steady: index*sr - L >= 0)
    t0 = P.window_start_sim(len(src), sr, 4)
    out, sg = hs.run([src], bank, [L], [dict(sound=0, t0=t0, rir=0)], sr, sr, fuse=True, simple=False)
    ref = O.compute_audiogoal(src, rir_wav, sr, audio_index=4)
    check(out[0], ref)
    check(sg[0], O.compute_spectrogram(ref.astype(np.float32)))


def test_spectrogram_kernel_pad_modes_and_edges():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 2, 16000)).astype(np.float32)
    x[1, :, :300] = 0.0
    for pm, name in ((0, "reflect"), (1, "constant")):
        got = hs.spectrogram(x, pad_mode=pm)
        for n in range(2):
            check(got[n], O.compute_spectrogram(x[n], pad_mode=name))
    assert hs.spectrogram(np.ones((1, 2, 16000), np.float32)).shape == (1, 65, 26, 2)   # nav.py:77 KAT


@pytest.mark.parametrize("n,gpw", [(16000, 3), (16000, 7), (44100, 4), (15999, 2), (4801, 1), (257, 1), (7919, 9)])
def test_spectrogram_kernel_chunked_rounds_and_odd_lengths(n, gpw):
    # one workgroup walks `gpw` groups of 4 pooled blocks (prefetching the next segment); odd lengths take the
    # scalar staging path, short rows have reflect padding on both sides inside one segment
    rng = np.random.default_rng(n)
    x = rng.standard_normal((2, 2, n)).astype(np.float32)
    for pm, name in ((0, "reflect"), (1, "constant")):
        got = hs.spectrogram(x, pad_mode=pm, gpw=gpw)
        assert got.shape[1:] == (65,) + (P.spectrogram_shape(n)[1], 2)
        for k in range(2):
            check(got[k], O.compute_spectrogram(x[k], pad_mode=name))


def test_window_planning():
    # 1-s clip at 16 kHz: only m = 0 is non-zero
    ws = P.plan_window_set(16000, 0, 1, 1)
    assert (ws.m_min, ws.count, ws.starts) == (0, 1, (-P.KB,))
    # 44.1 kHz: three RIR blocks x three output blocks, m in 0..2 (negative offsets are all-zero windows)
    ws = P.plan_window_set(44100, 0, 3, 3)
    assert (ws.m_min, ws.count) == (0, 3)
    # source exhausted -> nothing to do
    assert P.plan_window_set(16000, 5 * P.KB, 1, 1).count == 0
    row = P.unit_desc_row()
    assert row[0] == -1 and row[4] == -1


def test_intensity_kernel_vs_reference_run():
    z = golden()[0]
    a = z["clip1s/audiogoal"]
    rng = np.random.default_rng(2)
    late = np.zeros((2, 16000), np.float32)
    late[:, 15950:] = rng.standard_normal((2, 50)).astype(np.float32)            # onset near the end: < 150 samples
    x = np.stack([a, np.zeros_like(a), -np.abs(a) - 1.0, late])
    got = hs.intensity(x)
    np.testing.assert_allclose(got[0], z["clip1s/intensity"][0], rtol=2e-6)
    for n in range(4):
        np.testing.assert_allclose(got[n], O.intensity(x[n])[0], rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("seed", list(range(16)))
def test_randomised_edge_cases(seed):
    """Ragged / odd RIR lengths (down to 1 tap), odd n_valid, windows that run past the end of the clip, multi-second
    t0, both bank layouts, fused and unfused, SIMPLE and loop kernels — against the direct O(L*T) formula."""
    rng = np.random.default_rng(100 + seed)
    sr = 16000
    seconds = int(rng.integers(1, 4))
    src = O.synth_sources(rng, sr, k=1, seconds=seconds)[0]
    L = int(rng.choice([1, 2, 3, 777, 4001, 15999, 16000]))
    h = (O.synth_rir(rng, sr, length=L, n=1)[0] if L > 1000 else
         (0.3 * rng.standard_normal((2, L))).astype(np.float32))                # planar [2, L]
    cap = int(rng.choice([L + (L & 1), 16000, 16384]))
    cap = max(cap, L + (L & 1), 2)
    bank = np.zeros((1, 2, cap), np.float32)
    bank[0, :, :L] = h
    n_valid = int(rng.choice([sr, 4000, 4001, 1, 15999]))
    t0 = int(rng.choice([0, 5, sr, len(src) - 100, len(src) + 7]))
    kw = dict(fuse=bool(seed & 1), interleaved=bool(seed & 2) and not (seed & 1), simple=bool(rng.integers(0, 2)))
    out, sg = hs.run([src], bank, [L], [dict(sound=0, t0=t0, rir=0)], n_valid, sr, want_spectrogram=True, **kw)
    ref = np.zeros((2, sr))
    ref[:, :n_valid] = O.conv_window_direct(src, np.ascontiguousarray(h.T), t0, n_valid) if n_valid <= 512 else \
        O.conv_window_fft(src.astype(np.float64), np.ascontiguousarray(h.T).astype(np.float64), t0, n_valid)
    bound = np.abs(src).max() * np.abs(h).sum(axis=1).max()          # |out| <= max|x| * sum|h|: FFT rounding scales with it
    assert np.abs(out[0] - ref).max() <= 2e-6 * bound
    assert not out[0][:, n_valid:].any()
    if np.abs(ref).max() > 1e-3 * bound:
        check(sg[0], O.compute_spectrogram(ref.astype(np.float32)), tol=1e-4)
    if not kw["fuse"] and not kw["interleaved"] and not (cap & 1) and cap <= P.KB:
        # the persistent row kernel on the same ragged / odd-sized case (3 copies of the unit: rows walk 2 workgroups)
        out3, _ = hs.run([src], bank, [L], [dict(sound=0, t0=t0, rir=0)] * 3, n_valid, sr, persist=2)
        for k in range(3):
            np.testing.assert_array_equal(out3[k], out[0])


# ---- k_obs_rows: fused observation for rows of 2-3 partition blocks (44.1 kHz) -------------------------------------------
@pytest.mark.parametrize("spectral,stash", [(False, True), (True, False)])
def test_fused_rows_44k_vs_reference_vectors(spectral, stash):
    """One launch at the reference's Replica rate: simulator.py:629-632 on 44100-sample rows + nav.py:86-100 -> (65, 69, 2);
    the waveform is optional.  Time-domain bank (every RIR block transformed once per row, its spectrum stashed for the
    later output blocks) and spectral bank."""
    d = case_inputs("clip1s_44k")
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs("clip1s_44k")
    units = [dict(sound=0, t0=0, rir=0), dict(rir=-1)]
    out, sg = hs.run([d["source"]], planar(d["rir"]), [sr], units, sr, sr, row_wgs=1, spectral=spectral, row_stash=stash)
    assert sg.shape[1:] == (65, 69, 2)
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)
    assert not out[1].any() and not sg[1].any()                                  # silent unit: exact zeros
    none, sg2 = hs.run([d["source"]], planar(d["rir"]), [sr], units, sr, sr, row_wgs=4, spectral=spectral,
                       want_audiogoal=False)                                     # SpectrogramSensor alone: no waveform
    assert none is None
    np.testing.assert_array_equal(sg2, sg)


@pytest.mark.parametrize("wgs", [1, 3, 64])
def test_fused_rows_equal_two_kernel_path(wgs):
    """k_obs_rows against k_conv (loop kernel, forward FFTs re-run per output block) + k_spectrogram on the same inputs:
    multi-second clips in the early and the steady branch (negative partition offsets: several new RIR blocks in output
    block 0, all of them from the stash afterwards), a 2-s RIR (6 blocks), ragged and empty RIRs, a distractor, silent
    units in the middle of a walk."""
    rng = np.random.default_rng(23)
    sr = 44100
    srcs = [O.synth_sources(rng, sr, k=1, seconds=s)[0] for s in (1, 3, 1)]
    lens = [sr, 2 * sr, 0, 30000, 9001]
    cap = 2 * sr
    bank = np.zeros((len(lens), 2, cap), np.float32)
    for i, L in enumerate(lens):
        if L:
            bank[i, :, :L] = O.synth_rir(rng, sr, length=L, n=1)[0]
    units = [dict(sound=0, t0=0, rir=0),
             dict(sound=1, t0=P.window_start_sim(3 * sr, sr, 2), rir=1),         # steady branch, RIR longer than the row
             dict(rir=-1),
             dict(sound=1, t0=P.window_start_sim(3 * sr, sr, 1), rir=3),         # early branch
             dict(sound=0, t0=0, rir=2),                                         # empty RIR file: zero row
             dict(sound=2, t0=0, rir=4, dis_sound=0, dis_t0=0, dis_rir=3),       # distractor (simulator.py:649-664)
             dict(sound=1, t0=0, rir=0)]
    a_ref, s_ref = hs.run(srcs, bank, lens, units, sr, sr, fuse=False, want_spectrogram=True)
    a, sg = hs.run(srcs, bank, lens, units, sr, sr, row_wgs=wgs)
    # (the products of an output block are summed in a different order than in the loop kernel: equal to rounding)
    assert np.abs(a - a_ref).max() <= 2e-6 * np.abs(a_ref).max()
    assert np.abs(sg - s_ref).max() <= 1e-6 * np.abs(s_ref).max()
    assert not a[2].any() and not sg[2].any() and not a[4].any() and not sg[4].any()
    # and against the oracle, unit by unit
    check(a[1], O.compute_audiogoal(srcs[1], np.ascontiguousarray(bank[1].T), sr, audio_index=2))
    check(sg[1], O.compute_spectrogram(a_ref[1]))
    ref5 = O.compute_audiogoal(srcs[2], np.ascontiguousarray(bank[4, :, :9001].T), sr, distractor=srcs[0],
                               distractor_rir=np.ascontiguousarray(bank[3, :, :30000].T))
    check(a[5], ref5)
    # spectral bank, same units
    a_s, s_s = hs.run(srcs, bank, lens, units, sr, sr, row_wgs=wgs, spectral=True)
    assert np.abs(a_s - a_ref).max() <= 2e-6 * np.abs(a_ref).max()
    assert np.abs(s_s - s_ref).max() <= 2e-6 * np.abs(s_ref).max()


@pytest.mark.parametrize("out_len,n_valid,pad", [(44100, 11025, 0), (20000, 20000, 1), (32768, 32768, 0), (48000, 48000, 0),
                                                  (16386, 16386, 0), (44100, 0, 0)])
def test_fused_rows_lengths_pads_and_short_steps(out_len, n_valid, pad):
    """Row lengths around the block boundaries (2 and 3 blocks, a last block of 2 samples, 27 pooled blocks behind one
    output block), librosa >= 0.10 zero padding, SS2.0 0.25-s steps at 44.1 kHz (n_valid < out_len: blocks beyond
    n_valid are zeros), n_valid = 0; wav-interleaved bank rows."""
    rng = np.random.default_rng(out_len + n_valid)
    src = O.synth_sources(rng, out_len, k=1, seconds=2)[0]
    L = 25001
    bank = np.zeros((1, 2, L + 1), np.float32)
    bank[0, :, :L] = O.synth_rir(rng, out_len, length=L, n=1)[0]
    units = [dict(sound=0, t0=777, rir=0, wrap=False)]
    a_ref, s_ref = hs.run([src], bank, [L], units, n_valid, out_len, fuse=False, want_spectrogram=True, pad_mode=pad)
    a, sg = hs.run([src], bank, [L], units, n_valid, out_len, row_wgs=2, pad_mode=pad, interleaved=True)
    assert np.abs(a - a_ref).max() <= 2e-6 * max(1e-30, np.abs(a_ref).max())
    assert np.abs(sg - s_ref).max() <= 1e-6 * max(1e-30, np.abs(s_ref).max())
    assert not a[0, :, n_valid:].any()
    check(sg[0], O.compute_spectrogram(a_ref[0], pad_mode="constant" if pad else "reflect"))


# ---- length-bucketed RIR bank -----------------------------------------------------------------------------
@pytest.mark.parametrize("sr,fused_rows", [(16000, False), (44100, True)])
def test_bucketed_bank_short_and_long_rirs_in_one_launch(sr, fused_rows):
    """Bank entries 0..2 in a short bucket (rows of <= 1 block), entries 3..4 in a long bucket of its own (3-s RIRs: 3 / 9
    blocks); one launch mixes them (multi-second clips, a distractor from the other bucket, a silent unit).  Against the
    same units rendered from ONE bank at the long capacity, and the oracle."""
    rng = np.random.default_rng(sr)
    srcs = [O.synth_sources(rng, sr, k=1, seconds=s)[0] for s in (1, 4)]
    short_len, long_len = [int(0.4 * sr), int(0.25 * sr), 7000], [3 * sr, int(2.2 * sr)]
    cap_s, cap_l = max(short_len) + (max(short_len) & 1), 3 * sr
    b0 = np.zeros((3, 2, cap_s), np.float32)
    b1 = np.zeros((2, 2, cap_l), np.float32)
    for i, L in enumerate(short_len):
        b0[i, :, :L] = O.synth_rir(rng, sr, length=L, n=1)[0]
    for i, L in enumerate(long_len):
        b1[i, :, :L] = O.synth_rir(rng, sr, length=L, n=1)[0] * np.exp(-np.arange(L) / (0.8 * sr))[None, :].astype(np.float32)
    lens = short_len + long_len
    one = np.zeros((5, 2, cap_l), np.float32)
    one[:3, :, :cap_s] = b0
    one[3:] = b1
    units = [dict(sound=0, t0=0, rir=0),
             dict(sound=1, t0=P.window_start_sim(4 * sr, sr, 3), rir=3),          # steady branch through a 3-s RIR
             dict(rir=-1),
             dict(sound=1, t0=P.window_start_sim(4 * sr, sr, 1), rir=4, dis_sound=0, dis_t0=0, dis_rir=1),
             dict(sound=0, t0=0, rir=2, dis_sound=0, dis_t0=0, dis_rir=4)]
    kw = dict(row_wgs=3) if fused_rows else dict(fuse=True, simple=False)
    a_ref, s_ref = hs.run(srcs, one, lens, units, sr, sr, **kw)
    a, sg = hs.run(srcs, b0, lens, units, sr, sr, bucket2=b1, **kw)
    np.testing.assert_array_equal(a, a_ref)                                      # (same kernel, same order: same bits)
    np.testing.assert_array_equal(sg, s_ref)
    check(a[1], O.compute_audiogoal(srcs[1], np.ascontiguousarray(b1[0].T), sr, audio_index=3))
    check(a[4], O.compute_audiogoal(srcs[0], np.ascontiguousarray(b0[2, :, :7000].T), sr, distractor=srcs[0],
                                    distractor_rir=np.ascontiguousarray(b1[1, :, :long_len[1]].T)))
    assert not a[2].any() and not sg[2].any()
    if not fused_rows:                                   # a launch that stays in bucket 0 keeps the loop-free kernel
        u0 = [dict(sound=0, t0=0, rir=i) for i in range(3)]
        a0, s0 = hs.run(srcs, b0, lens, u0, sr, sr, fuse=True, bucket2=b1)
        a1, s1 = hs.run(srcs, b0, short_len, u0, sr, sr, fuse=True)
        np.testing.assert_array_equal(a0, a1)
        np.testing.assert_array_equal(s0, s1)


SIM_44K = [c for c in golden()[1] if c.endswith("_44k") and c.startswith(("clip1s_ragged", "multi_", "distractor"))]


@pytest.mark.parametrize("name", SIM_44K)
def test_fused_rows_vs_reference_run_vectors_at_44k(name):
    """k_obs_rows against vectors produced by RUNNING the reference's _compute_audiogoal at 44.1 kHz (make_golden.py, round
    3): 3-s clips in the early and the steady branch (all three RIR blocks new in output block 0, everything from the
    stash afterwards), a 1.5-s RIR (5 blocks), a ragged RIR, a distractor."""
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    t0 = P.window_start_sim(len(d["source"]), sr, d.get("audio_index", 0))
    srcs, banks, lens = [d["source"]], [d["rir"]], [d["rir"].shape[0]]
    u = dict(sound=0, t0=t0, rir=0)
    if "distractor" in d:
        srcs.append(d["distractor"]); banks.append(d["distractor_rir"]); lens.append(d["distractor_rir"].shape[0])
        u.update(dis_sound=1, dis_t0=0, dis_rir=1)
    cap = max(lens) + (max(lens) & 1)
    bank = np.concatenate([planar(b, cap) for b in banks])
    out, sg = hs.run(srcs, bank, lens, [u], sr, sr, row_wgs=2)
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)
    _, sg_s = hs.run(srcs, bank, lens, [u], sr, sr, row_wgs=2, spectral=True, want_audiogoal=False)
    check(sg_s[0], ref_s)


@pytest.mark.parametrize("name", ["cont_early_44k", "cont_steady_44k"])
def test_fused_rows_continuous_steps_at_44k(name):
    """SS2.0 0.25-s steps at 44.1 kHz (reference-run: continuous_simulator.py:413-456): 11025 valid samples of a
    44100-sample row - one convolved block, two blocks of zeros, the spectrogram over the whole row."""
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    src3 = O.tile_short_source(d["source"], sr)
    ns = int(sr * d["step_time"])
    wrap = d["sample_index"] - d["rir"].shape[0] >= 0
    out, sg = hs.run([src3], planar(d["rir"]), [d["rir"].shape[0]],
                     [dict(sound=0, t0=P.window_start_continuous(d["sample_index"]), rir=0, wrap=wrap)], ns, sr, row_wgs=1)
    assert not out[0][:, ns:].any()
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)


def test_fused_rows_four_second_rir_eleven_blocks():
    """SS2.0's ray-traced RIRs run to 4 s: 176400 taps = 11 partition blocks at 44.1 kHz.  A 5-s clip in the steady branch
    (output block 0 finds 11 new RIR blocks: ten stash-only forward FFTs, the eleventh carries the products; blocks 1, 2 are
    served from the stash alone) and a 0.25-s SS2.0 step with wrap-around, against the oracle."""
    rng = np.random.default_rng(41)
    sr, L = 44100, 4 * 44100
    src = O.synth_sources(rng, sr, k=1, seconds=5)[0]
    h = O.synth_rir(rng, sr, length=L, n=1)[0] * np.exp(-np.arange(L) / (0.9 * sr))[None, :].astype(np.float32)
    bank = h[None].astype(np.float32)
    rir_wav = np.ascontiguousarray(h.T)
    t0 = P.window_start_sim(len(src), sr, 4)
    out, sg = hs.run([src], bank, [L], [dict(sound=0, t0=t0, rir=0)], sr, sr, row_wgs=1)
    ref = O.compute_audiogoal(src, rir_wav, sr, audio_index=4)
    check(out[0], ref)
    check(sg[0], O.compute_spectrogram(ref.astype(np.float32)))
    ns = sr // 4
    out, sg = hs.run([src], bank, [L], [dict(sound=0, t0=P.window_start_continuous(200000), rir=0, wrap=True)], ns, sr,
                     row_wgs=1)
    ref = O.convolve_with_rir(src, rir_wav, sr, 200000, 0.25)
    check(out[0], ref)
    check(sg[0], O.compute_spectrogram(ref.astype(np.float32)))


@pytest.mark.parametrize("step_time,len_prev,sample_index", [(0.25, 20000, 50000), (0.25, 40000, 9000), (1.0, 30000, 70000)])
def test_fused_rows_crossfade_at_44k(step_time, len_prev, sample_index):
    """SS_FLAG_CROSSFADE in k_obs_rows (continuous_simulator.py:47-53, 413-426 at the reference's Replica rate): block 0 is
    rendered twice - previous RIR (term 1), then current - and blended over int(0.05 sr) + 1 samples; a 1-s step (three
    rendered blocks) keeps the previous RIR out of blocks 1-2; a 40000-tap previous RIR (3 RIR blocks, early branch: it is
    longer than the sample index) against a 20000-tap current one (steady); a unit without a previous RIR in the same launch."""
    sr = 44100
    rng = np.random.default_rng(77)
    src3 = O.tile_short_source(O.synth_sources(rng, sr, k=1, seconds=1)[0], sr)
    cur = np.ascontiguousarray(O.synth_rir(rng, sr, length=20000, n=1)[0].T)
    prev = np.ascontiguousarray(O.synth_rir(rng, sr, length=len_prev, n=1)[0].T)
    ns = int(sr * step_time)
    cap = max(20000, len_prev)
    bank = np.concatenate([planar(cur, cap), planar(prev, cap)])
    wrap_cur, wrap_prev = sample_index - 20000 >= 0, sample_index - len_prev >= 0
    t0 = P.window_start_continuous(sample_index)
    units = [dict(sound=0, t0=t0, rir=0, wrap=wrap_cur, last_rir=1, last_wrap=wrap_prev), dict(sound=0, t0=t0, rir=0, wrap=wrap_cur)]
    out, sg = hs.run([src3], bank, [20000, len_prev], units, ns, sr, crossfade=True, row_wgs=3, want_spectrogram=True)
    ref = O.compute_audiogoal_continuous(src3, cur, sr, sample_index, step_time, last_rir=prev, use_crossfade=True)
    plain = O.convolve_with_rir(src3, cur, sr, sample_index, step_time)
    check(out[0], ref)
    check(out[1], plain)
    check(sg[0], O.compute_spectrogram(ref.astype(np.float32)))
    check(sg[1], O.compute_spectrogram(plain.astype(np.float32)))
    n = int(0.05 * sr)
    assert np.abs(out[0][:, n + 1:] - out[1][:, n + 1:]).max() == 0.0          # beyond the ramp: the current RIR alone
    assert np.abs(out[0][:, :n] - out[1][:, :n]).max() > 0.0
    assert not out[0][:, ns:].any()


def test_fused_rows_crossfade_44k_reference_run_vector():
    """cont_crossfade_44k: produced by running the reference's ContinuousSoundSpacesSim._compute_audiogoal with CROSSFADE
    on at 44.1 kHz (tests/golden/make_golden.py)."""
    d = case_inputs("cont_crossfade_44k")
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs("cont_crossfade_44k")
    src3 = O.tile_short_source(d["source"], sr)
    ns = int(sr * d["step_time"])
    bank = np.concatenate([planar(d["rir"]), planar(d["last_rir"])])
    t0 = P.window_start_continuous(d["sample_index"])
    out, sg = hs.run([src3], bank, [d["rir"].shape[0], d["last_rir"].shape[0]],
                     [dict(sound=0, t0=t0, rir=0, wrap=True, last_rir=1)], ns, sr, crossfade=True, row_wgs=2)
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)


@pytest.mark.parametrize("sr,n_valids", [(16000, [1, 383, 384, 385, 1024, 4000, 15487, 15488, 15489, 15999]),
                                         (44100, [11025, 16384, 16385, 27000, 43588, 43589])])
def test_zero_pooled_blocks_are_written_not_computed_at_any_step_length(sr, n_valids):
    """Steps shorter than the row (n_valid < out_len): the pooled blocks behind the rendered samples are exact zeros and the
    kernels write them without running their STFT - on both sides of every boundary of that rule (640 b - 256 >= n_valid;
    right padding mirrors zeros only while n_valid <= out_len - 512), fused 16 kHz kernel and k_obs_rows."""
    rng = np.random.default_rng(5)
    src3 = O.tile_short_source(O.synth_sources(rng, sr, k=1, seconds=1)[0], sr)
    rir = np.ascontiguousarray(O.synth_rir(rng, sr, length=9000, n=1)[0].T)
    idx = 20000
    full = O.convolve_with_rir(src3, rir, sr, idx, 1.0)                    # steady branch, no wrap: idx + sr < 3 sr
    t0 = P.window_start_continuous(idx)
    for nv in n_valids:
        ref = full.copy()
        ref[:, nv:] = 0
        ref_s = O.compute_spectrogram(ref.astype(np.float32))
        # fused kernel of the rate | the two-launch formulation (loop kernel, then k_spectrogram told where the zeros begin)
        for kw in ((dict(row_wgs=2) if sr > P.KB else dict(fuse=True)), dict(fuse=False, simple=False)):
            out, sg = hs.run([src3], planar(rir), [9000], [dict(sound=0, t0=t0, rir=0, wrap=True)], nv, sr,
                             want_spectrogram=True, **kw)
            check(out[0], ref)
            check(sg[0], ref_s)
            assert (sg[0][ref_s == 0] == 0).all()                         # the written zeros are exact


# ---- k_conv<FUSE, loop, XFADE?, WIDE>: one launch for rows of which only block 0 is rendered (SS2.0 steps at 44.1 kHz) ------
@pytest.mark.parametrize("name", ["cont_early_44k", "cont_steady_44k", "cont_crossfade_44k", "cont_early_48k", "cont_steady_48k",
                                  "cont_crossfade_48k"])
def test_wide_one_block_kernel_vs_reference_run_vectors(name):
    """The reference's ContinuousSoundSpacesSim._compute_audiogoal at 44.1 kHz (continuous_simulator.py:413-456, CROSSFADE
    :47-53): a 0.25-s step renders 11025 samples of a 44100-sample row - block 0 only.  The fused LOOP kernel serves it in
    one launch (block spectra accumulated in registers, 18 live pooled columns, 51 written as zeros)."""
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    src3 = O.tile_short_source(d["source"], sr)
    ns = int(sr * d["step_time"])
    xf = "last_rir" in d
    if xf:
        bank = np.concatenate([planar(d["rir"]), planar(d["last_rir"])])
        lens = [d["rir"].shape[0], d["last_rir"].shape[0]]
        unit = dict(sound=0, t0=P.window_start_continuous(d["sample_index"]), rir=0, wrap=True, last_rir=1)   # (both steady)
    else:
        bank, lens = planar(d["rir"]), [d["rir"].shape[0]]
        unit = dict(sound=0, t0=P.window_start_continuous(d["sample_index"]), rir=0,
                    wrap=d["sample_index"] - d["rir"].shape[0] >= 0)
    out, sg = hs.run([src3], bank, lens, [unit], ns, sr, fuse=True, simple=False, crossfade=xf)
    assert not out[0][:, ns:].any()
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)


@pytest.mark.parametrize("len_prev,sample_index", [(20000, 50000), (40000, 9000)])
def test_wide_one_block_crossfade_equals_oracle_and_rows_kernel(len_prev, sample_index):
    sr = 44100
    rng = np.random.default_rng(78)
    src3 = O.tile_short_source(O.synth_sources(rng, sr, k=1, seconds=1)[0], sr)
    cur = np.ascontiguousarray(O.synth_rir(rng, sr, length=20000, n=1)[0].T)
    prev = np.ascontiguousarray(O.synth_rir(rng, sr, length=len_prev, n=1)[0].T)
    ns = sr // 4
    cap = max(20000, len_prev)
    bank = np.concatenate([planar(cur, cap), planar(prev, cap)])
    wrap_cur, wrap_prev = sample_index - 20000 >= 0, sample_index - len_prev >= 0
    t0 = P.window_start_continuous(sample_index)
    units = [dict(sound=0, t0=t0, rir=0, wrap=wrap_cur, last_rir=1, last_wrap=wrap_prev), dict(sound=0, t0=t0, rir=0, wrap=wrap_cur),
             dict(rir=-1)]
    out, sg = hs.run([src3], bank, [20000, len_prev], units, ns, sr, crossfade=True, fuse=True, simple=False)
    ref = O.compute_audiogoal_continuous(src3, cur, sr, sample_index, 0.25, last_rir=prev, use_crossfade=True)
    plain = O.convolve_with_rir(src3, cur, sr, sample_index, 0.25)
    check(out[0], ref)
    check(out[1], plain)
    check(sg[0], O.compute_spectrogram(ref.astype(np.float32)))
    check(sg[1], O.compute_spectrogram(plain.astype(np.float32)))
    assert not out[2].any() and not sg[2].any()
    out_r, sg_r = hs.run([src3], bank, [20000, len_prev], units, ns, sr, crossfade=True, row_wgs=2)
    assert np.abs(out - out_r).max() <= 2e-6 * np.abs(out_r).max() and np.abs(sg - sg_r).max() <= 2e-6 * sg_r.max()


def test_wide_one_block_every_step_length_up_to_one_block():
    """n_valid from 1 sample to a full block (16384: 26 live pooled columns, the most the fused phase holds), a distractor
    term, both pad modes: the WIDE kernel against the oracle, zero columns exact."""
    sr = 44100
    rng = np.random.default_rng(6)
    srcs = [O.tile_short_source(s, sr) for s in O.synth_sources(rng, sr, k=2, seconds=1)]
    rirs = O.synth_rir(rng, sr, length=30000, n=2)
    bank = np.ascontiguousarray(rirs)
    idx = 20000
    a = O.convolve_with_rir(srcs[0], np.ascontiguousarray(rirs[0].T), sr, idx, 1.0)
    b = O.convolve_with_rir(srcs[1], np.ascontiguousarray(rirs[1].T), sr, idx, 1.0)
    t0 = P.window_start_continuous(idx)
    for nv, pad in ((1, 0), (383, 1), (11025, 0), (11025, 1), (15999, 0), (16384, 0)):
        units = [dict(sound=0, t0=t0, rir=0, wrap=False), dict(sound=0, t0=t0, rir=0, wrap=False, dis_sound=1, dis_t0=t0, dis_rir=1)]
        out, sg = hs.run(srcs, bank, [30000, 30000], units, nv, sr, fuse=True, simple=False, pad_mode=pad)
        for n, ref in enumerate((a, a + b)):
            ref = ref.copy()
            ref[:, nv:] = 0
            ref_s = O.compute_spectrogram(ref.astype(np.float32), pad_mode=("reflect", "constant")[pad])
            check(out[n], ref)
            check(sg[n], ref_s)
            assert (sg[n][ref_s == 0] == 0).all()


def test_fused_rows_48k_reference_run_vector():
    """sim48k_multi_i1: the reference's _compute_audiogoal at 48 kHz (3-s clip, second 1: the steady branch hears the tail of
    second 0) - three blocks per row through k_obs_rows."""
    d = case_inputs("sim48k_multi_i1")
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs("sim48k_multi_i1")
    t0 = P.window_start_sim(len(d["source"]), sr, d["audio_index"])
    out, sg = hs.run([d["source"]], planar(d["rir"]), [d["rir"].shape[0]], [dict(sound=0, t0=t0, rir=0)], sr, sr, row_wgs=2)
    check(out[0][:, ::stride], ref_a)
    check(sg[0], ref_s)


# ---- the 512-thread / 32-values-per-thread core (ss_fft_core32.hpp, ss_kernels32.hpp) ---------------------------------
@pytest.mark.parametrize("name", SIM_CASES)
def test_core32_sim_branches_vs_reference_vectors(name):
    d = case_inputs(name)
    sr = d["sr"]
    if d["rir"].shape[0] > P.KB:
        pytest.skip("the loop-free kernel serves RIRs of one block")
    ref_a, ref_s, stride = case_outputs(name)
    t0 = P.window_start_sim(len(d["source"]), sr, d.get("audio_index", 0))
    for tab in (False, True):
        out, sg = hs.run([d["source"]], planar(d["rir"]), [d["rir"].shape[0]], [dict(sound=0, t0=t0, rir=0)],
                         sr, sr, fuse=True, core32=True, tab=tab)
        check(out[0][:, ::stride], ref_a)
        check(sg[0], ref_s)


def test_core32_equals_1024_thread_core_on_a_mixed_batch():
    # silent unit, empty RIR, ragged lengths, interleaved (wav) bank layout, a 0.25-s step, fused and unfused
    rng = np.random.default_rng(11)
    sr = 16000
    srcs = O.synth_sources(rng, sr, k=3)
    bank = np.zeros((4, 2, sr), np.float32)
    lens = [sr, 5000, 0, 12345]
    for i, L in enumerate(lens):
        if L:
            bank[i, :, :L] = O.synth_rir(rng, sr, length=L, n=1)[0]
    units = [dict(sound=0, t0=0, rir=0), dict(rir=-1), dict(sound=1, t0=0, rir=1), dict(sound=2, t0=0, rir=2),
             dict(sound=1, t0=0, rir=3)]
    for n_valid in (sr, 4000):
        a_ref, s_ref = hs.run(srcs, bank, lens, units, n_valid, sr, fuse=True)
        for kw in (dict(fuse=True), dict(fuse=True, tab=True), dict(fuse=False, want_spectrogram=True),
                   dict(fuse=True, interleaved=True), dict(fuse=True, pad_mode=1)):
            if kw.get("pad_mode"):
                a_ref2, s_ref2 = hs.run(srcs, bank, lens, units, n_valid, sr, fuse=True, pad_mode=1)
            else:
                a_ref2, s_ref2 = a_ref, s_ref
            a, sg = hs.run(srcs, bank, lens, units, n_valid, sr, core32=True, **kw)
            assert O.relerr(a, a_ref2) <= 2e-6, O.relerr(a, a_ref2)
            assert O.relerr(sg, s_ref2) <= 2e-6, O.relerr(sg, s_ref2)
            assert not a[1].any() and not a[3].any()          # silent unit / empty RIR: exact zeros
            assert not a[:, :, n_valid:].any()


@pytest.mark.parametrize("parts_log2", [1, 2, 3])
@pytest.mark.parametrize("variant", ["simple", "loop", "spectral", "short_step"])
def test_split_rows_equal_one_workgroup_per_row(parts_log2, variant):
    """Small steps: a fused one-block row rendered by 2^k workgroups (ConvParams::parts_log2: the whole convolution in each,
    the pooled STFT blocks shared out) is BIT-identical to the one-workgroup row: the audiogoal comes from part 0, every
    pooled column from exactly one part."""
    d = case_inputs("clip1s_ragged")
    sr = d["sr"]
    bank = planar(d["rir"])
    units = [dict(sound=0, t0=0, rir=0), dict(sound=0, t0=0, rir=-1), dict(sound=0, t0=0, rir=0)]     # (one silent unit)
    n_valid = 4000 if variant == "short_step" else sr
    kw = dict(fuse=True, simple=variant != "loop", spectral=variant == "spectral")
    a1, s1 = hs.run([d["source"]], bank, [d["rir"].shape[0]], units, n_valid, sr, **kw)
    a2, s2 = hs.run([d["source"]], bank, [d["rir"].shape[0]], units, n_valid, sr, parts_log2=parts_log2, **kw)
    np.testing.assert_array_equal(a1, a2)
    np.testing.assert_array_equal(s1, s2)
    if variant != "short_step":
        _, ref_s, _ = case_outputs("clip1s_ragged")
        check(s2[0], ref_s)
        assert not s2[1].any()


@pytest.mark.parametrize("parts_log2", [1, 3])
@pytest.mark.parametrize("variant", ["time", "spectral", "short_step", "crossfade"])
def test_split_rows_of_the_44k_kernel_equal_one_workgroup_per_row(parts_log2, variant):
    """k_obs_rows with a row on 2 / 8 workgroups (small steps at the reference's Replica rate - 5 envs per GPU at 44.1 kHz,
    ss_baselines/av_nav/config/audionav/replica/train_telephone/audiogoal_depth_ddppo.yaml:3 +
    configs/audionav/av_nav/replica/audiogoal.yaml:18): every part renders the row's three blocks, the pooled STFT blocks of
    each phase are shared out.  Bit-identical to the one-workgroup row, incl. a silent unit, a distractor, a 0.25-s step
    (zero columns) and a cross-faded row."""
    d = case_inputs("clip1s_44k")
    sr = d["sr"]
    rng = np.random.default_rng(5)
    bank = np.concatenate([planar(d["rir"]), planar(np.ascontiguousarray(O.synth_rir(rng, sr, length=d["rir"].shape[0], n=1)[0].T))])
    lens = [d["rir"].shape[0]] * 2
    units = [dict(sound=0, t0=0, rir=0), dict(sound=0, t0=0, rir=-1), dict(sound=0, t0=0, rir=1, dis_sound=0, dis_t0=0, dis_rir=0)]
    kw = dict(fuse=True, row_wgs=64, spectral=variant == "spectral")
    n_valid = 11025 if variant == "short_step" else sr
    src = [O.tile_short_source(d["source"], sr)] if variant in ("short_step", "crossfade") else [d["source"]]
    if variant == "crossfade":
        units = [dict(sound=0, t0=9000, rir=0, last_rir=1), dict(sound=0, t0=9000, rir=1)]
        kw.update(crossfade=True)
    a1, s1 = hs.run(src, bank, lens, units, n_valid, sr, **kw)
    a2, s2 = hs.run(src, bank, lens, units, n_valid, sr, parts_log2=parts_log2, **kw)
    np.testing.assert_array_equal(a1, a2)
    np.testing.assert_array_equal(s1, s2)
    if variant == "time":
        ref_a, ref_s, stride = case_outputs("clip1s_44k")
        check(s2[0], ref_s)
        assert not s2[1].any()


@pytest.mark.parametrize("parts_log2", [0, 2])
@pytest.mark.parametrize("variant", ["time", "spectral", "multi_second", "long_rir"])
def test_one_workgroup_per_output_block_equals_the_row_kernel(parts_log2, variant):
    """k_obs_blocks (round 6: small steps at the reference's Replica rate - one workgroup per OUTPUT BLOCK of a row, the samples
    behind a block boundary handed to the next block's workgroup through global memory and a flag) against k_obs_rows on the same
    launch: the same arithmetic per output sample and pooled column up to the order of a block's sum - time-domain bank (forward transforms
    accumulated in registers, no stash) and spectral bank, a silent unit, a distractor, a 3-s clip in the steady branch
    (simulator.py:641-647: pairs from several windows), a 1.5-s RIR (five RIR blocks), rows split over parts."""
    name = {"multi_second": "multi_L1.0_i2_44k", "long_rir": "multi_L1.5_i2_44k"}.get(variant, "clip1s_44k")
    d = case_inputs(name)
    sr = d["sr"]
    rng = np.random.default_rng(5)
    L = d["rir"].shape[0]
    bank = np.concatenate([planar(d["rir"]), planar(np.ascontiguousarray(O.synth_rir(rng, sr, length=L, n=1)[0].T))])
    lens = [L] * 2
    t0 = P.window_start_sim(len(d["source"]), sr, d.get("audio_index", 0))
    short = O.synth_sources(np.random.default_rng(9), sr, k=1)[0]
    units = [dict(sound=0, t0=t0, rir=0), dict(sound=0, t0=0, rir=-1), dict(sound=0, t0=t0, rir=1, dis_sound=1, dis_t0=0, dis_rir=0)]
    kw = dict(fuse=True, row_wgs=64, spectral=variant == "spectral")
    a1, s1 = hs.run([d["source"], short], bank, lens, units, sr, sr, **kw)
    a2, s2 = hs.run([d["source"], short], bank, lens, units, sr, sr, parts_log2=parts_log2, row_blocks=True, **kw)
    for n in (0, 2):                                       # (the products of a block are summed in another order: equal to rounding)
        assert O.relerr(s2[n], s1[n]) < 2e-6 and O.relerr(a2[n], a1[n]) < 2e-6
    ref_a, ref_s, stride = case_outputs(name)
    check(s2[0], ref_s)
    check(a2[0][:, ::stride], ref_a)
    assert not s2[1].any() and not a2[1].any()


@pytest.mark.parametrize("sr", [22050, 32000, 48000])
def test_one_workgroup_per_output_block_at_other_rates(sr):
    """k_obs_blocks on rows of two blocks (22.05 kHz), of exactly two blocks' worth of pooled columns (32 kHz: the last block
    completes a single pooled block more than the first) and of three blocks at 48 kHz: against the oracle (simulator.py:629-632
    + nav.py:86-100) and against k_obs_rows."""
    rng = np.random.default_rng(sr)
    src = O.synth_sources(rng, sr, k=1)[0]
    h = O.synth_rir(rng, sr, n=2)
    bank = np.ascontiguousarray(h)
    units = [dict(sound=0, t0=0, rir=0), dict(sound=0, t0=0, rir=1)]
    kw = dict(fuse=True, row_wgs=64)
    a1, s1 = hs.run([src], bank, [sr, sr], units, sr, sr, **kw)
    a2, s2 = hs.run([src], bank, [sr, sr], units, sr, sr, row_blocks=True, parts_log2=1, **kw)
    for n in range(2):
        ref = O.compute_audiogoal(src, np.ascontiguousarray(h[n].T), sr)
        check(a2[n], ref)
        check(s2[n], O.compute_spectrogram(ref.astype(np.float32)))
        assert O.relerr(s2[n], s1[n]) < 2e-6 and O.relerr(a2[n], a1[n]) < 2e-6


def test_a_hand_off_that_never_arrives_poisons_the_block_s_first_column():
    """k_obs_blocks waits (bounded) for the previous block's workgroup to leave the samples behind the block boundary.  The launcher
    only takes the kernel when every workgroup is resident, so the wait cannot run out - if it ever did, the block's first frame
    is rendered from NaN, not from stale samples: a loud observation instead of a plausible wrong one.  Host build, workgroups in
    REVERSE order (no hand-off has been written when its reader runs): one NaN pooled column per hand-off, everything else -
    the waveform, the other columns - as in the ordered run."""
    sr = 44100
    rng = np.random.default_rng(3)
    src = O.synth_sources(rng, sr, k=1)[0]
    bank = np.ascontiguousarray(O.synth_rir(rng, sr, n=1))
    units = [dict(sound=0, t0=0, rir=0)]
    kw = dict(fuse=True, row_wgs=64)
    for spectral in (False, True):
        a1, s1 = hs.run([src], bank, [sr], units, sr, sr, row_blocks=True, spectral=spectral, **kw)
        a2, s2 = hs.run([src], bank, [sr], units, sr, sr, row_blocks=2, spectral=spectral, **kw)
        assert not np.isnan(s1).any() and np.array_equal(a1, a2)
        bad = np.isnan(s2[0])                                     # [65, T4, 2]
        cols = np.flatnonzero(bad.any(axis=(0, 2)))
        assert len(cols) == 2 and bad[:, cols, :].all()           # three output blocks: two hand-offs, whole columns, both ears
        assert np.array_equal(s2[0][~bad], s1[0][~bad])

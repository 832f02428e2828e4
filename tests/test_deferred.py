"""Deferred mode (ss_amd/deferred.py): env workers in separate PROCESSES (habitat.VectorEnv, the reference's default,
ss_baselines/common/env_utils.py:91-107) ship AudioRequests through pipes; the trainer process renders all envs in one
launch.  CPU only: oracle-backed engine (tests/fakes.py); the GPU half is test_gpu_parity.py::test_deferred_on_gpu."""
import multiprocessing as mp
import pickle

import numpy as np
import pytest
import torch

from oracle import ss_oracle as O
from fakes import FakeContinuousSim, FakeSim, NS, OracleEngine
from ss_amd import sensors, sim_audio
from ss_amd.deferred import AudioRequest, DeferredResolver, attach_deferred

SR = 16000


def make_world(seed=3):
    rng = np.random.default_rng(seed)
    sounds = {"telephone.wav": O.synth_sources(rng, SR, k=1)[0], "long.wav": O.synth_sources(rng, SR, k=1, seconds=3)[0],
              "dist.wav": O.synth_sources(rng, SR, k=1)[0]}
    files = {}
    for az in (0, 90, 180, 270):
        for r in range(4):
            for s in (7, 11):
                files[f"rirs/replica/apartment_0/{az}/{r}_{s}.wav"] = np.ascontiguousarray(
                    O.synth_rir(rng, SR, length=int(rng.integers(900, 2500)), n=1)[0].T)
    return sounds, files


def trajectory(rank, steps):
    rng = np.random.default_rng(100 + rank)
    return [(int(rng.integers(0, 4)), int(rng.integers(0, 4)) * 90, "long.wav" if (rank + k // 3) % 2 else "telephone.wav")
            for k in range(steps)]


def apply(sim, k, move):
    recv, rot, sound = move
    if sim._current_sound != sound:
        sim._current_sound, sim._audio_index = sound, 0
    sim._receiver_position_index, sim._rotation_angle = recv, rot
    sim._episode_step_count = k
    sim._duration = 6


def worker(rank, conn, steps, has_distractor):
    sounds, files = make_world()
    sim = FakeSim(SR, sounds, files, has_distractor)
    sim._current_distractor_sound = "dist.wav"
    attach_deferred(sim, env_rank=rank)
    sg = sensors.SpectrogramSensor(sim=sim, config=NS())
    ag = sensors.AudioGoalSensor(sim=sim, config=NS())
    for k, move in enumerate(trajectory(rank, steps)):
        apply(sim, k, move)
        obs = {"spectrogram": sg.get_observation(observations=None, episode=None),
               "audiogoal": ag.get_observation(observations=None, episode=None), "depth": np.zeros((2, 2), np.float32)}
        conn.send(obs)                                    # pickled, like habitat.VectorEnv's pipe
    conn.close()


@pytest.mark.parametrize("has_distractor", [False, True])
def test_two_worker_processes_one_launch_per_step(has_distractor):
    steps, n_env = 9, 2
    ctx = mp.get_context("fork")
    pipes, procs = [], []
    for rank in range(n_env):
        a, b = ctx.Pipe()
        p = ctx.Process(target=worker, args=(rank, b, steps, has_distractor), daemon=True)
        p.start()
        b.close()
        pipes.append(a)
        procs.append(p)
    sounds, files = make_world()
    eng = OracleEngine(SR)
    resolver = DeferredResolver(eng, rir_reader=files.get)
    # the reference path, per env, in this process
    twins = [FakeSim(SR, sounds, files, has_distractor) for _ in range(n_env)]
    ref_eng = OracleEngine(SR)
    backs = []
    for t in twins:
        t._current_distractor_sound = "dist.wav"
        backs.append(sim_audio.HipSimAudio(t, ref_eng, rir_reader=files.get))
    trajs = [trajectory(r, steps) for r in range(n_env)]
    try:
        _drive(steps, n_env, pipes, trajs, eng, resolver, twins, backs, ref_eng)
    finally:
        for p in procs:                                   # a failing assertion must not leave workers blocked on send()
            p.join(5)
            if p.is_alive():
                p.terminate()
    for p in procs:
        assert p.exitcode == 0


def _drive(steps, n_env, pipes, trajs, eng, resolver, twins, backs, ref_eng):
    for k in range(steps):
        for pipe in pipes:
            assert pipe.poll(60), "worker process died or hung"
        observations = [pipe.recv() for pipe in pipes]
        assert all(isinstance(o["spectrogram"], AudioRequest) for o in observations)
        assert observations[0]["spectrogram"].clip is None or k == 0 or observations[0]["spectrogram"].sound != trajs[0][k - 1][2]
        q0 = observations[0]["spectrogram"]
        clips = sum(4 * len(c) for c in (q0.clip, q0.dis_clip) if c is not None)
        assert len(pickle.dumps(q0)) < clips + 1000                                # a few hundred bytes + first-use clips
        calls = eng.calls
        # ONE launch for the vector step - unless the two sensors of a step name different clip windows: with
        # HAS_DISTRACTOR_SOUND the reference caches nothing (simulator.py:679-681), every sensor read computes again and
        # advances `_audio_index` again (multi-second sounds, :634-635): one launch per sensor then
        two_windows = any(o["audiogoal"].t0 != o["spectrogram"].t0 for o in observations)
        batch = resolver.resolve_observations(observations)
        assert eng.calls == calls + (2 if two_windows else 1)
        assert tuple(batch["spectrogram"].shape) == (n_env, 65, 26, 2) and tuple(batch["audiogoal"].shape) == (n_env, 2, SR)
        assert torch.equal(observations[1]["spectrogram"], batch["spectrogram"][1])   # dicts now hold tensors
        for r, (t, b) in enumerate(zip(twins, backs)):
            apply(t, k, trajs[r][k])
            req = b.unit_request()                                                 # the SpectrogramSensor's read ...
            want = ref_eng.observe([req], want_audiogoal=True)
            assert torch.allclose(batch["spectrogram"][r], want["spectrogram"][0], atol=1e-6)
            if t.config.AUDIO.HAS_DISTRACTOR_SOUND:                               # ... and the AudioGoalSensor's: computed AGAIN
                want = ref_eng.observe([b.unit_request()], want_audiogoal=True)
            assert torch.allclose(batch["audiogoal"][r], want["audiogoal"][0], atol=1e-6)
    assert not batch["audiogoal"].any()                                            # steps 7, 8 > duration 6: silent


def test_deferred_continuous_requests_carry_the_live_rirs():
    rng = np.random.default_rng(4)
    sounds = {"telephone": O.synth_sources(rng, SR, k=1)[0]}
    bank = O.synth_rir(rng, SR, length=9000, n=8)
    sims = [FakeContinuousSim(SR, sounds, lambda k, o=o: bank[(k + o) % 8].astype(np.float64).tolist(), start_index=500 * o)
            for o in range(3)]
    workers = [attach_deferred(s, env_rank=i, continuous=True) for i, s in enumerate(sims)]
    eng = OracleEngine(SR, step_time=0.25)
    resolver = DeferredResolver(eng)
    for step in range(6):
        reqs = [pickle.loads(pickle.dumps(s.get_current_spectrogram_observation(None))) for s in sims]
        assert reqs[0].live_rir.shape == (9000, 2) and (reqs[0].last_rir is None) == (step == 0)
        out = resolver.resolve(reqs, want_audiogoal=True)
        for i, s in enumerate(sims):
            assert O.relerr(out["audiogoal"][i].numpy(), s.reference_audiogoal()) < 1e-5
        for s in sims:
            s.step()
    assert eng.uploads == 3 * 6                       # one new RIR per env and step (the previous one is recognised)
    assert workers[0].request("audiogoal") is workers[0].request("spectrogram")     # one request per simulator state

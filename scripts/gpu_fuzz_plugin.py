"""Randomised parity sweep at the PLUGIN boundary (test infrastructure; the oracle is the checker): simulators that walk.

Every trial builds a scene of RIR wav files on tmpfs (float32 files of ragged lengths, one int16 file, one empty file, one file
that is not a wav), a bank of sounds (1-s and multi-second clips) and E stand-in simulators (`tests/fakes.py::FakeSim`: the
attributes SoundSpacesSim._compute_audiogoal reads, simulator.py:608-666) that walk for a number of steps - receiver / source
moves, rotations, new episodes (another sound, `_audio_index` back to 0, a short `_duration` so that the episode turns silent,
:610-612), with or without a distractor (:649-664).  The same walk is served three ways and every observation of every step is
compared with the oracle at the north-star tolerance:

  eager     `sim_audio.attach` + the task sensors (`SpectrogramSensor` / `AudioGoalSensor`), one simulator at a time
  deferred  `attach_deferred` on the worker side (requests pickled as through habitat.VectorEnv's pipe), `DeferredResolver` on
            the trainer side (column path, misses served inside the step's C call), a store SMALLER than the scene
  batched   `sim_audio.VectorAudioObserver` over the eager adapters (in-process vector envs)

and `_audio_index` must have advanced exactly as the reference advances it (:634-635) - once per COMPUTATION: without a distractor
the second sensor of a step hits the per-pose memo (:682-686), with HAS_DISTRACTOR_SOUND nothing is cached (:679-681) and the two
sensors of a step hear consecutive seconds of a multi-second sound (eager and deferred: two sensor reads per step; the batched
observer is ONE read per step by construction).

    python scripts/gpu_fuzz_plugin.py --trials 40 --seed 1 [--out profiles/r6/fuzz/plugin_seed1.txt]
"""
import argparse
import os
import pickle
import shutil
import sys
import tempfile
import time

import numpy as np
import torch
from scipy.io import wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd"), os.path.join(ROOT, "tests")]

from fakes import FakeSim, NS                          # noqa: E402
from oracle import ss_oracle as O                      # noqa: E402
from ss_amd import sensors, sim_audio                  # noqa: E402
from ss_amd.deferred import DeferredResolver, attach_deferred   # noqa: E402
from ss_amd.renderer import AudioEngine                # noqa: E402

TOL = 1e-4
DEV = "cuda:0"


def build_scene(rng, sr, root, n_nodes):
    """-> {path: [L, 2] float32 array or None (the reference's zero-RIR fallback, simulator.py:619-624)}"""
    rirs = {}
    odd = {(0, 0, 1): "i16", (90, 1, 0): "empty", (180, 1, 1): "junk"}
    for az in (0, 90, 180, 270):
        os.makedirs(os.path.join(root, str(az)))
        for r in range(n_nodes):
            for s in range(n_nodes):
                p = os.path.join(root, str(az), f"{r}_{s}.wav")
                kind = odd.get((az, r, s))
                L = int(rng.uniform(0.03, 1.0) * sr) if rng.random() < 0.85 else int(rng.uniform(1.0, 1.5) * sr)
                h = np.ascontiguousarray(O.synth_rir(rng, sr, length=L, n=1)[0].T).astype(np.float32)
                if kind == "i16":
                    # (integer PCM is convolved as it is, simulator.py:615-618.  A modest scale: at 20000 x the spectrogram's log1p
                    #  turns logarithmic in the quiet frames of a decaying RIR and shows the float32 convolution floor - 1e-6 of the
                    #  row's peak, the reference's own scipy float32 path has the same - as 1.4e-4 of the spectrogram's peak)
                    q = (h * 300).astype(np.int16)
                    wavfile.write(p, sr, q)
                    rirs[p] = q.astype(np.float32)
                elif kind == "empty":
                    wavfile.write(p, sr, np.zeros((0, 2), np.float32))
                    rirs[p] = None
                elif kind == "junk":
                    open(p, "wb").write(b"RIFFthis is not a wav file")
                    rirs[p] = None
                else:
                    wavfile.write(p, sr, h)
                    rirs[p] = h
    return rirs


def make_sims(sr, sounds, root, n_env, has_distractor):
    sims = []
    for _ in range(n_env):
        sim = FakeSim(sr, sounds, {}, has_distractor=has_distractor)
        sim.binaural_rir_dir = root
        sims.append(sim)
    return sims


def apply_state(sim, st, keep_caches=False):
    (sim._current_sound, sim._audio_index, sim._episode_step_count, sim._duration, sim._receiver_position_index,
     sim._source_position_index, sim._distractor_position_index, sim._rotation_angle, sim._current_distractor_sound) = st
    if not keep_caches:                                # (memo trials keep them: the reference's per-pose caches, :678-701)
        sim._audiogoal_cache.clear()
        sim._spectrogram_cache.clear()


def run_trial(rng, base):
    sr = int(rng.choice([16000, 16000, 16000, 44100]))
    n_nodes = int(rng.integers(2, 6 if sr == 16000 else 4))
    n_env = int(rng.choice([1, 2, 3, 5, 8, 16] if sr == 16000 else [1, 2, 5]))
    has_dis = bool(rng.random() < 0.35)
    root = tempfile.mkdtemp(dir=base)
    try:
        rirs = build_scene(rng, sr, root, n_nodes)
        sounds = {}
        for k in range(int(rng.integers(2, 5))):
            n = sr if rng.random() < 0.5 else int(rng.integers(2, 5)) * sr
            sounds[f"s{k}.wav"] = (rng.standard_normal(n) * rng.uniform(0.05, 0.5)).astype(np.float32)
        names = list(sounds)
        one_s = [n_ for n_ in names if len(sounds[n_]) == sr] or names
        n_files = len(rirs)
        lo = max(4, 2 * n_env * (2 if has_dis else 1))
        slots = int(rng.integers(lo, max(lo + 1, n_files + 8)))
        sets = {m: make_sims(sr, sounds, root, n_env, has_dis) for m in ("eager", "deferred", "batched")}
        eng_e = AudioEngine(sr, device=DEV, rir_slots=slots)
        eng_d = AudioEngine(sr, device=DEV, rir_slots=slots)
        eng_b = AudioEngine(sr, device=DEV, rir_slots=slots)
        lazy = bool(rng.random() < 0.5)
        # memo trials: the simulators keep their per-pose caches between steps (simulator.py:682-686, 694-698: a pose seen before
        # returns what was first rendered there, `_audio_index` untouched; new dicts when an episode changes the sound, :395-397);
        # eager mode uses the simulator's own dicts, deferred mode `pose_cache=True`; the batched observer renders every step
        memo = (not has_dis) and bool(rng.random() < 0.5)
        model = {m: [dict() for _ in range(n_env)] for m in ("eager", "deferred")}
        for sim in sets["eager"]:
            sim_audio.attach(sim, eng_e, lazy_audiogoal=lazy)
        for sim in sets["batched"]:
            sim_audio.attach(sim, eng_b)
        for i, sim in enumerate(sets["deferred"]):
            attach_deferred(sim, env_rank=i, pose_cache=memo)
        res = DeferredResolver(eng_d, fast=True, prefetch_azimuths=bool(rng.random() < 0.3))
        sg_s = [sensors.SpectrogramSensor(sim=s_, config=NS()) for s_ in sets["eager"]]
        ag_s = [sensors.AudioGoalSensor(sim=s_, config=NS()) for s_ in sets["eager"]]
        observer = sim_audio.VectorAudioObserver(eng_b, [s_._ss_hip_audio for s_ in sets["batched"]], want_audiogoal=True)
        # walk state per env: (sound, audio index, step count, duration, receiver, source, distractor node, rotation, distractor sound)
        state = [[names[0], 0, 0, 500, 0, 1 % n_nodes, 0, 0, one_s[0] if has_dis else None] for _ in range(n_env)]
        worst, n_obs = 0.0, 0
        history = [[] for _ in range(n_env)]
        index = {m: [0] * n_env for m in sets}            # `_audio_index` per serving mode (they read a different number of times)
        for step in range(int(rng.integers(4, 12))):
            for e in range(n_env):
                st = state[e]
                if rng.random() < 0.15:                               # a new episode
                    prev = st[0]
                    # (trials that clear the simulators' caches before every step - the cache-miss path - keep the step count
                    #  running: a simulator never computes twice in one state, and the deferred adapter relies on that when it
                    #  hands the second sensor of a step the first one's request)
                    st[0], st[1], st[2] = str(rng.choice(names)), 0, 0 if memo else st[2]
                    for m in index:
                        index[m][e] = 0
                    if memo and st[0] != prev:                        # another sound: the reference starts new caches (:395-397)
                        for m in model:
                            model[m][e] = dict()
                            sets[m][e]._audiogoal_cache, sets[m][e]._spectrogram_cache = dict(), dict()
                    st[3] = st[2] + int(rng.choice([2, 3, 500]))
                    if has_dis:
                        st[8], st[6] = str(rng.choice(one_s)), int(rng.integers(0, n_nodes))
                if rng.random() < 0.8:
                    st[4] = int(rng.integers(0, n_nodes))
                if rng.random() < 0.2:
                    st[5] = int(rng.integers(0, n_nodes))
                st[7] = int(rng.choice([0, 90, 180, 270]))
                st[2] += 1
            def reference(e, idx):
                """(audiogoal, spectrogram, _audio_index afterwards) of ONE computation (simulator.py:608-666) from clip second idx"""
                snd, _, cnt, dur, recv, src, dnode, rot, dsnd = state[e]
                az = -(rot + 0) % 360
                silent = cnt > dur
                h = rirs[os.path.join(root, str(az), f"{recv}_{src}.wav")]
                hd = rirs[os.path.join(root, str(az), f"{recv}_{dnode}.wav")] if has_dis else None
                a = O.compute_audiogoal(sounds[snd], h if h is not None else O.zero_rir(sr), sr, idx, silent,
                                        sounds[dsnd] if has_dis else None,
                                        (hd if hd is not None else O.zero_rir(sr)) if has_dis else None)
                a = np.asarray(a, np.float64)
                nxt = idx if (silent or len(sounds[snd]) == sr) else (idx + 1) % (len(sounds[snd]) // sr)
                return a, O.compute_spectrogram(a.astype(np.float32)), nxt

            def expect(e, idx, reads, mode=None):
                """spectrogram read first, then (reads == 2) the audiogoal read: the memo serves it without a distractor"""
                if memo and mode in model:
                    pose = (state[e][5], state[e][4], -(state[e][7] + 0) % 360)       # (source, receiver, azimuth), :683
                    if pose not in model[mode][e]:
                        a1, s1, n1 = reference(e, idx)
                        model[mode][e][pose] = (a1, s1)
                        return a1, s1, n1
                    return model[mode][e][pose] + (idx,)
                a1, s1, n1 = reference(e, idx)
                if reads == 1 or not has_dis:
                    return a1, s1, n1
                a2, _, n2 = reference(e, n1)
                return a2, s1, n2

            for e in range(n_env):
                history[e].append(tuple(state[e][:1] + state[e][2:6] + state[e][7:8]))
            got, want = {}, {}
            # eager: two sensor reads per simulator
            for e, sim in enumerate(sets["eager"]):
                apply_state(sim, tuple(state[e][:1] + [index["eager"][e]] + state[e][2:]), memo)
            out = []
            for e, sim in enumerate(sets["eager"]):
                s_ = sg_s[e].get_observation(observations=None, episode=None)
                a_ = ag_s[e].get_observation(observations=None, episode=None)
                out.append((np.asarray(a_), np.asarray(s_), sim._audio_index))
            got["eager"] = out
            want["eager"] = [expect(e, index["eager"][e], 2, "eager") for e in range(n_env)]
            # deferred: both sensors on the worker side, the dicts pickled as through the vector env's pipe
            for e, sim in enumerate(sets["deferred"]):
                apply_state(sim, tuple(state[e][:1] + [index["deferred"][e]] + state[e][2:]), memo)
            observations = [pickle.loads(pickle.dumps({"spectrogram": sim.get_current_spectrogram_observation(None),
                                                       "audiogoal": sim.get_current_audiogoal_observation()}))
                            for sim in sets["deferred"]]
            o = res.resolve_observations(observations, replace=False)
            ag, sg = o["audiogoal"].cpu().numpy(), o["spectrogram"].cpu().numpy()
            got["deferred"] = [(ag[e], sg[e], sets["deferred"][e]._audio_index) for e in range(n_env)]
            want["deferred"] = [expect(e, index["deferred"][e], 2, "deferred") for e in range(n_env)]
            # batched: one read per simulator and step
            for e, sim in enumerate(sets["batched"]):
                apply_state(sim, tuple(state[e][:1] + [index["batched"][e]] + state[e][2:]))
            o = observer.observe()
            ag, sg = o["audiogoal"].cpu().numpy(), o["spectrogram"].cpu().numpy()
            got["batched"] = [(ag[e], sg[e], sets["batched"][e]._audio_index) for e in range(n_env)]
            want["batched"] = [expect(e, index["batched"][e], 1) for e in range(n_env)]
            for mode, outs in got.items():
                for e, (a, s_, nxt) in enumerate(outs):
                    ra, rs, rn = want[mode][e]
                    assert nxt == rn, f"step {step} env {e} {mode}: _audio_index {nxt} != {rn} state={state[e]} from {index[mode][e]} memo={memo} " \
                                      f"history={history[e]}"
                    for g, r, what in ((a, ra, "audiogoal"), (s_, rs, "spectrogram")):
                        assert g.shape == r.shape, f"{mode} {what} shape {g.shape} != {r.shape}"
                        assert not np.isnan(g).any(), f"step {step} env {e} {mode} {what}: NaN"
                        scale = np.abs(r).max()
                        err = float(np.abs(g - r).max() / scale) if scale > 0 else float(np.abs(g).max())
                        worst = max(worst, err)
                        assert err <= TOL, f"step {step} env {e} {mode} {what}: {err:.3e} state={state[e]} from {index[mode][e]} sr={sr} slots={slots}"
                    n_obs += 1
                    index[mode][e] = rn
        return sr, n_env, n_nodes, slots, has_dis, n_obs, worst
    finally:
        shutil.rmtree(root, ignore_errors=True)


def run_continuous_trial(rng, base):
    """SoundSpaces 2.0 simulators (continuous_simulator.py:370-462; `tests/fakes.py::FakeContinuousSim` steps them the way the
    reference's `step` does, :384-390): a live RIR from the ray tracer every step (ragged lengths, also longer than a second), the
    previous step's RIR for the cross-fade (:422-424), a sample index that wraps around the clip (:438-447), episodes that restart
    (`_last_rir` None, a new random sample index, :341-342) or run past `_duration`.  Eager (`attach_continuous` + the task sensors),
    deferred (`attach_deferred(continuous=True)`: numbered live RIRs through the pipe) and batched (`VectorAudioObserver`)."""
    from fakes import FakeContinuousSim
    sr = int(rng.choice([16000, 16000, 44100, 48000]))
    step_time = float(rng.choice([0.25, 0.25, 0.1]))
    crossfade = bool(rng.random() < 0.7)
    n_env = int(rng.choice([1, 2, 3, 5, 8] if sr == 16000 else [1, 2, 4]))
    sounds = {}
    for k in range(int(rng.integers(1, 4))):
        n = sr if rng.random() < 0.5 else int(rng.uniform(2.0, 4.0) * sr)
        sounds[f"s{k}.wav"] = (rng.standard_normal(n) * rng.uniform(0.05, 0.5)).astype(np.float32)
    names = list(sounds)
    pools = []
    for e in range(n_env):
        pool = []
        for _ in range(int(rng.integers(3, 7))):
            L = int(rng.uniform(0.05, 1.0) * sr) if rng.random() < 0.85 else int(rng.uniform(1.0, 1.4) * sr)
            pool.append(O.synth_rir(rng, sr, length=L, n=1)[0].astype(np.float64))          # [2, L], as the audio sensor returns it
        pools.append(pool)
    starts = [int(rng.integers(0, int(sr * step_time))) for _ in range(n_env)]

    def make(e):
        pool = pools[e]
        return FakeContinuousSim(sr, sounds, lambda k, pool=pool: pool[(3 * k + 1) % len(pool)].tolist(), step_time=step_time,
                                 crossfade=crossfade, start_index=starts[e])
    modes = ("eager", "deferred", "batched")
    sets = {m: [make(e) for e in range(n_env)] for m in modes}
    engs = {m: AudioEngine(sr, device=DEV, rir_slots=2 * n_env + 4, step_time=step_time, wrap=True) for m in modes}
    for sim in sets["eager"]:
        sim_audio.attach_continuous(sim, engs["eager"])
    for sim in sets["batched"]:
        sim_audio.attach_continuous(sim, engs["batched"])
    for i, sim in enumerate(sets["deferred"]):
        attach_deferred(sim, env_rank=i, continuous=True)
    res = DeferredResolver(engs["deferred"])
    sg_s = [sensors.SpectrogramSensor(sim=s_, config=NS()) for s_ in sets["eager"]]
    ag_s = [sensors.AudioGoalSensor(sim=s_, config=NS()) for s_ in sets["eager"]]
    observer = sim_audio.VectorAudioObserver(engs["batched"], [s_._ss_hip_audio for s_ in sets["batched"]], want_audiogoal=True)
    worst, n_obs = 0.0, 0
    for step in range(int(rng.integers(5, 14))):
        refs = [sets["eager"][e].reference_audiogoal() for e in range(n_env)]
        refs = [(np.asarray(a, np.float64), O.compute_spectrogram(np.asarray(a, np.float32))) for a in refs]
        got = {}
        got["eager"] = [(np.asarray(ag_s[e].get_observation(observations=None, episode=None)),
                         np.asarray(sg_s[e].get_observation(observations=None, episode=None))) for e in range(n_env)]
        observations = [pickle.loads(pickle.dumps({"spectrogram": sim.get_current_spectrogram_observation(None),
                                                   "audiogoal": sim.get_current_audiogoal_observation()}))
                        for sim in sets["deferred"]]
        o = res.resolve_observations(observations, replace=False)
        ag, sg = o["audiogoal"].cpu().numpy(), o["spectrogram"].cpu().numpy()
        got["deferred"] = [(ag[e], sg[e]) for e in range(n_env)]
        o = observer.observe()
        ag, sg = o["audiogoal"].cpu().numpy(), o["spectrogram"].cpu().numpy()
        got["batched"] = [(ag[e], sg[e]) for e in range(n_env)]
        for mode, outs in got.items():
            for e, (a, s_) in enumerate(outs):
                for g, r, what in ((a, refs[e][0], "audiogoal"), (s_, refs[e][1], "spectrogram")):
                    assert g.shape == r.shape, f"{mode} {what} shape {g.shape} != {r.shape}"
                    assert not np.isnan(g).any(), f"step {step} env {e} {mode} {what}: NaN"
                    scale = np.abs(r).max()
                    err = float(np.abs(g - r).max() / scale) if scale > 0 else float(np.abs(g).max())
                    worst = max(worst, err)
                    sim = sets[mode][e]
                    assert err <= TOL, f"step {step} env {e} {mode} {what}: {err:.3e} sr={sr} step_time={step_time} crossfade={crossfade} " \
                                       f"index={sim._current_sample_index} clip={sim.current_source_sound.shape[0]} count={sim._episode_step_count}"
                n_obs += 1
        for e in range(n_env):                                    # the same transition on the three twins
            restart = rng.random() < 0.12
            snd = str(rng.choice(names))
            idx = int(rng.integers(0, int(sr * step_time)))
            dur = int(rng.choice([1, 2, 500]))
            for m in modes:
                sim = sets[m][e]
                sim.step()
                if restart:                                       # reconfigure (:336-346): new episode
                    sim._current_sound, sim._last_rir, sim._current_sample_index = snd, None, idx
                    sim._episode_step_count, sim._duration = 0, dur
    return sr, n_env, 0, 2 * n_env + 4, crossfade, n_obs, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", type=int, default=-1, help="run this one trial (debugging)")
    ap.add_argument("--mode", choices=["sim", "continuous"], default="sim", help="SoundSpacesSim walks / SoundSpaces 2.0 walks")
    args = ap.parse_args()
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    lines, fails, worst_all, n_all = [], 0, 0.0, 0
    t_start = time.time()
    for t in (range(args.trials) if args.only < 0 else [args.only]):
        rng = np.random.default_rng([args.seed, t])
        try:
            sr, n_env, n_nodes, slots, dis, n_obs, worst = (run_trial if args.mode == 'sim' else run_continuous_trial)(rng, base)
            worst_all, n_all = max(worst_all, worst), n_all + n_obs
            lines.append(f"trial {t:4d} ok   sr={sr:5d} envs={n_env:2d} nodes={n_nodes} store={slots:3d} distractor={int(dis)} "
                         f"observations={n_obs:4d} worst={worst:.2e}")
        except Exception as e:                          # noqa: BLE001 - a sweep reports every failing trial
            fails += 1
            lines.append(f"trial {t:4d} FAIL {type(e).__name__}: {e}")
        print(lines[-1], flush=True)
    tail = f"# plugin boundary, {args.mode} (eager / deferred / batched), {args.trials} trials, seed {args.seed}: {fails} failed, {n_all} observations, " \
           f"worst relative error {worst_all:.2e} (tolerance {TOL:.0e}), {time.time() - t_start:.0f} s"
    print(tail)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines + [tail]) + "\n")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()

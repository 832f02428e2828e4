#!/usr/bin/env python
"""Deferred mode for SoundSpaces 2.0 (ContinuousSoundSpacesSim, the reference's default DD-PPO mode, in the reference's default
multi-process arrangement): every worker-side sensor returns an AudioRequest carrying the step's LIVE RIR ([L, 2] float32 from
the ray tracer) and the previous one (CROSSFADE); the trainer resolves the N requests in one launch.  Times the trainer half
(request walk, RIR uploads - one pinned block per step since round 4 -, planner, launch) per vector step.
usage: bench_deferred_continuous.py [--envs 128] [--rir-len 9000] [--steps 60]"""
import argparse, json, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from oracle import ss_oracle as O
from ss_amd.deferred import DeferredResolver, attach_deferred
from ss_amd.renderer import AudioEngine
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=128)
ap.add_argument("--rir-len", type=int, default=9000)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--profile", action="store_true", help="cProfile of the trainer half over the timed steps")
ap.add_argument("--scatter-copy", action="store_true", help="A/B: H2D copy of the staged rows, then the scatter on the device copy "
                "(default: the scatter kernel reads the pinned block itself)")
ap.add_argument("--gather-threads", type=int, default=0, help="A/B: host threads of the row gather (default: the store's 8)")
ap.add_argument("--breakdown", action="store_true", help="perf_counter around the stages of the trainer half")
ap.add_argument("--walk", action="store_true", help="force the per-request walk (round 4's path: fast=False) - same-box A/B")
a = ap.parse_args()
sr, N, L = 16000, a.envs, a.rir_len
NS = types.SimpleNamespace
rng = np.random.default_rng(0)
clip = O.tile_short_source(O.synth_sources(rng, sr, k=1)[0], sr)
pool = O.synth_rir(rng, sr, length=L, n=32)                       # [32, 2, L]: what the ray tracer hands over ([2][L])


class Sim:
    """continuous_simulator.py:370-462: the attributes its audio code reads, step() advancing them like :384-390"""

    def __init__(self, o):
        self.config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, CROSSFADE=True), STEP_TIME=0.25)
        self._source_sound_dict = {"s": clip}
        self._current_sound, self._episode_step_count, self._duration = "s", 0, 10 ** 9
        self._o, self._k = o, 0
        self._prev_sim_obs = {"audio_sensor": pool[o % 32]}
        self._last_rir, self._current_sample_index = None, 500 * o

    current_source_sound = property(lambda self: self._source_sound_dict[self._current_sound])

    def step(self):
        self._last_rir = np.transpose(np.array(self._prev_sim_obs["audio_sensor"]))
        self._k += 1
        self._prev_sim_obs = {"audio_sensor": pool[(self._o + self._k) % 32]}
        self._episode_step_count += 1
        self._current_sample_index = int(self._current_sample_index + sr * 0.25) % clip.shape[0]


sims = [Sim(o) for o in range(N)]
for i, s in enumerate(sims):
    attach_deferred(s, env_rank=i, continuous=True)
eng = AudioEngine(sr, device="cuda:0", rir_slots=2 * N + 8, rir_cap=L, step_time=0.25, wrap=True)
eng.store.scatter_from_host = not a.scatter_copy
if a.gather_threads > 0:
    eng.store.gather_threads = a.gather_threads
acc = {}
if a.breakdown:
    from ss_amd import renderer as _R, deferred as _D

    def _wrap(obj, name, tag):
        f = getattr(obj, name)

        def g(*ar, **kw):
            t = time.perf_counter()
            try:
                return f(*ar, **kw)
            finally:
                acc[tag] = acc.get(tag, 0) + time.perf_counter() - t
        setattr(obj, name, g)
    _wrap(_R.RirStore, "upload_rows", "upload_rows")
    _wrap(_R.RirStore, "_scatter_staged", "scatter_staged")
    _wrap(_R.AudioEngine, "observe_columns", "observe_columns")
    _wrap(_D.DeferredResolver, "_live_columns", "live_columns (incl. upload_rows)")
    _wrap(_D.DeferredResolver, "resolve", "resolve")
    from ss_amd import _lib as _L
    _wrap(_L.load(), "ss_rows_gather_f32", "ss_rows_gather_f32")
res = DeferredResolver(eng, fast=False) if a.walk else DeferredResolver(eng)
sg = torch.empty((N, 65, 26, 2), device="cuda:0")
w_us, t_us = [], []
import cProfile, pstats, io
pr = cProfile.Profile()
for k in range(a.warmup + a.steps):
    if k == a.warmup:
        torch.cuda.synchronize(); t_start = time.perf_counter()
        acc.clear()
    t0 = time.perf_counter()
    reqs = [s.get_current_spectrogram_observation(None) for s in sims]
    t1 = time.perf_counter()
    if a.profile and k >= a.warmup:
        pr.enable()
    res.resolve(reqs, spectrogram_out=sg)
    if a.profile and k >= a.warmup:
        pr.disable()
    t2 = time.perf_counter()
    for s in sims:
        s.step()
    if k >= a.warmup:
        w_us.append(1e6 * (t1 - t0)); t_us.append(1e6 * (t2 - t1))
torch.cuda.synchronize()
dt = time.perf_counter() - t_start
print(json.dumps({"mode": "deferred, SoundSpaces 2.0 live RIRs + CROSSFADE", "scatter": "copy + device scatter" if a.scatter_copy else "kernel reads the pinned block", "path": "request walk (r4)" if a.walk else "live columns (r5)",
                  "live_steps": res.live_steps, "walk_steps": res.walk_steps, "envs": N, "rir_len": L, "steps": a.steps,
                  "trainer_half_us_per_step": round(float(np.median(t_us)), 1),
                  "trainer_half_env_steps_per_s": round(N / (np.median(t_us) * 1e-6), 1),
                  "worker_half_us_per_step_all_envs": round(float(np.median(w_us)), 1),
                  "h2d_mb_per_step": round(N * 2 * L * 4 / 1e6, 2), "pcie_floor_us_per_step": round(N * 2 * L * 4 / 63e3, 1),
                  "both_halves_and_sim_steps_serial_env_steps_per_s": round(N * a.steps / dt, 1)}))
if a.breakdown:
    nst = a.steps
    print("breakdown, us per step (mean over the %d timed steps): " % nst + ", ".join("%s %.1f" % (k_, 1e6 * v / nst) for k_, v in acc.items()))
if a.profile:
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:5000])

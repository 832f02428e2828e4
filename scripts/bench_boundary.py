#!/usr/bin/env python
"""The boundary, mode by mode, measured at one shape: N stand-in simulators (the attributes
soundspaces/simulator.py's audio code reads), 16 kHz, 1-s clips, RIR 'files' served from memory and resident in the HBM
store after the first touch; per vector step every agent moves, then the step's spectrograms are produced
  eager      the reference's call chain, one env at a time: sensor -> sim.get_current_spectrogram_observation() ->
             batch-1 launch -> D2H -> numpy (pose cache defeated: the cache-miss path)
  batched    VectorAudioObserver.observe_into(rollouts): per-env Python unit_request(), one launch, rollout rows
  deferred   worker half (AudioRequest per env; runs inside the env processes in real use) + trainer half
             (DeferredResolver.resolve_observations -> one launch into the rollout rows), timed separately
  columns    FastVectorAudioObserver.observe_into(rollouts) (ss_ctx_observe_sims)
One JSON line per mode.  usage: bench_boundary.py [--envs 128] [--steps 200]"""
import argparse, json, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=128)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--nodes", type=int, default=8)
args = ap.parse_args()

from ss_amd import planning as P, sensors, sim_audio
from ss_amd.context import AudioContext
from ss_amd.deferred import DeferredResolver, attach_deferred
from ss_amd.renderer import AudioEngine
from ss_amd.rollout import RolloutStorage
from ss_amd.vector import FastVectorAudioObserver, RirIndex, VectorSimState

dev, sr, N, n_nodes = "cuda:0", 16000, args.envs, args.nodes
rng = np.random.default_rng(0)
NS = types.SimpleNamespace
t = np.arange(sr) / sr
sounds = {"sound%d" % i: (0.5 * np.sin(2 * np.pi * (200 + 37 * i) * t) * np.hanning(sr)).astype(np.float32) for i in range(16)}
env = np.exp(-6.9 * np.arange(sr) / (0.5 * sr)).astype(np.float32)[:, None]
files = {}
for r in range(n_nodes):
    for s in range(n_nodes):
        for az in (0, 90, 180, 270):
            files["rirs/%d/%d_%d.wav" % (az, r, s)] = (rng.standard_normal((sr, 2)).astype(np.float32) * env * 0.1)


class Sim:
    """soundspaces/simulator.py:110-117, 303-305, 568-573 (attributes) and :500-516 (what a step changes)."""

    def __init__(self):
        self.config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, HAS_DISTRACTOR_SOUND=False), USE_RENDERED_OBSERVATIONS=True)
        self._source_sound_dict = sounds
        self.binaural_rir_dir = "rirs"
        self._current_sound = "sound%d" % rng.integers(0, len(sounds))
        self._current_distractor_sound = None
        self._episode_step_count, self._duration = 0, 10 ** 9
        self._receiver_position_index = int(rng.integers(0, n_nodes))
        self._source_position_index = int(rng.integers(0, n_nodes))
        self._distractor_position_index = 0
        self._rotation_angle = int(rng.integers(0, 4)) * 90
        self._audio_index = 0
        self._audiogoal_cache, self._spectrogram_cache = {}, {}

    azimuth_angle = property(lambda self: -(self._rotation_angle + 0) % 360)
    current_source_sound = property(lambda self: self._source_sound_dict[self._current_sound])
    _audio_length = property(lambda self: self.current_source_sound.shape[0] // sr)

    def move(self, action, node):
        if action == 0:
            self._receiver_position_index = node
        elif action == 1:
            self._rotation_angle = (self._rotation_angle + 90) % 360
        else:
            self._rotation_angle = (self._rotation_angle - 90) % 360
        self._episode_step_count += 1
        self._spectrogram_cache.clear(); self._audiogoal_cache.clear()       # measure the cache-miss path


def rollouts_for(n):
    space = NS(spaces={"spectrogram": NS(shape=P.spectrogram_shape(sr))})

    class ActionSpace:
        pass
    return RolloutStorage(16, n, space, ActionSpace(), 8, device=dev)


total = args.warmup + args.steps
acts = rng.integers(0, 3, (total, N))
nodes = rng.integers(0, n_nodes, (total, N))


def timed(step_fn, sims):
    host = []
    for k in range(total):
        if k == args.warmup:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, sim in enumerate(sims):
            sim.move(acts[k, i], int(nodes[k, i]))
        h0 = time.perf_counter()
        step_fn(k)
        if k >= args.warmup:
            host.append(1e6 * (time.perf_counter() - h0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"env_steps_per_s": round(N * args.steps / dt, 1), "ms_per_step": round(1e3 * dt / args.steps, 4),
            "audio_host_us_per_step_median": round(float(np.median(host)), 1)}


reader = files.get
out = {}

# eager
eng = AudioEngine(sr, device=dev, rir_slots=4 * n_nodes * n_nodes, rir_spectral=True)
sims = [Sim() for _ in range(N)]
for s in sims:
    sim_audio.attach(s, eng, rir_reader=reader)
sens = [sensors.SpectrogramSensor(sim=s, config=NS()) for s in sims]
out["eager"] = timed(lambda k: [se.get_observation(observations=None, episode=None) for se in sens], sims)

# batched in-process
sims = [Sim() for _ in range(N)]
backs = [sim_audio.attach(s, eng, rir_reader=reader) for s in sims]
obs = sim_audio.VectorAudioObserver(eng, backs)
ro = rollouts_for(N)


def batched(k):
    obs.observe_into(ro)
    ro.step = (ro.step + 1) % 16
out["batched"] = timed(batched, sims)

# deferred: worker half and trainer half
sims = [Sim() for _ in range(N)]
defs = [attach_deferred(s, env_rank=i) for i, s in enumerate(sims)]
res = DeferredResolver(eng, rir_reader=reader)
ro = rollouts_for(N)
w_us, t_us = [], []


def deferred(k):
    a = time.perf_counter()
    observations = [{"spectrogram": s.get_current_spectrogram_observation()} for s in sims]
    b = time.perf_counter()
    res.resolve_observations(observations, ro)
    ro.step = (ro.step + 1) % 16
    c = time.perf_counter()
    if k >= args.warmup:
        w_us.append(1e6 * (b - a)); t_us.append(1e6 * (c - b))
out["deferred"] = timed(deferred, sims)
out["deferred"].update(worker_half_us_per_step_all_envs=round(float(np.median(w_us)), 1),
                       trainer_half_us_per_step=round(float(np.median(t_us)), 1),
                       note="worker half runs inside the env processes in real use (N-way parallel); the trainer pays the trainer half")

# columns
sims = [Sim() for _ in range(N)]
ctx = AudioContext(sr)

index = RirIndex(4)
sid = index.add_scene("synthetic", n_nodes)
store2_keys = [(r, s) for r in range(n_nodes) for s in range(n_nodes)]
from ss_amd.renderer import RirStore
st = RirStore(slots=4 * len(store2_keys), cap=sr, device=dev, group=4, spectral=True)
bases = st.slot_many(store2_keys, [(lambda r=r, s=s: [files["rirs/%d/%d_%d.wav" % (az, r, s)] for az in (0, 90, 180, 270)])
                                   for r, s in store2_keys])
for (r, s), b in zip(store2_keys, bases):
    index.set(sid, r, s, b)
st.sync_spectra()
ctx.set_rir_bank(st.bank.data, st.bank.lengths)
ctx.set_rir_spectra(st.bank.spectra)
state = VectorSimState(N)
for i, s in enumerate(sims):
    state.bind(s, i)
state.scene[:] = sid
fobs = FastVectorAudioObserver(ctx, state, index, sr)
ro = rollouts_for(N)


def columns(k):
    fobs.observe_into(ro)
    ro.step = (ro.step + 1) % 16
out["columns"] = timed(columns, sims)

for mode, v in out.items():
    print(json.dumps({"mode": mode, "envs": N, "steps": args.steps, **v}))

#!/usr/bin/env python
"""Soak of the in-call miss path (ss_ctx_observe_requests_load: reader + scatter + table update + eviction inside the step's call):
deferred mode over real wav files, a store far smaller than the scene (constant eviction), random walks, varying step sizes -
every step compared BIT FOR BIT with a resolver on the three-call path over a store that holds the whole scene.
usage: python scripts/gpu_soak_miss.py [seed] [steps]"""
import os, pickle, sys, tempfile, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from scipy.io import wavfile
from oracle import ss_oracle as O
from ss_amd.deferred import DeferredResolver, attach_deferred
from ss_amd.renderer import AudioEngine

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
NS = types.SimpleNamespace
sr, n_nodes, n_env = 16000, 7, 24
rng = np.random.default_rng(seed)
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
for az in (0, 90, 180, 270):
    os.makedirs(os.path.join(td, str(az)))
    for r in range(n_nodes):
        for s_ in range(n_nodes):
            L = int(rng.integers(1500, 16001))
            h = O.synth_rir(np.random.default_rng(1000 * az + 10 * r + s_), sr, length=L, n=1)[0]
            wavfile.write(os.path.join(td, str(az), f"{r}_{s_}.wav"), sr, np.ascontiguousarray(h.T))
clips = {f"s{i}.wav": c for i, c in enumerate(O.synth_sources(np.random.default_rng(5), sr, k=3))}


class Sim:
    config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, HAS_DISTRACTOR_SOUND=False), USE_RENDERED_OBSERVATIONS=True)
    binaural_rir_dir = td
    _source_sound_dict = clips
    _audio_index, _episode_step_count, _duration = 0, 0, 10 ** 9
    _audio_length = 1

    def __init__(self):
        self._current_sound = "s0.wav"
        self._receiver_position_index = self._source_position_index = 0
        self.azimuth_angle = 0

    @property
    def current_source_sound(self):
        return clips[self._current_sound]


def make(in_call, slots):
    sims = [Sim() for _ in range(n_env)]
    for i, sm in enumerate(sims):
        attach_deferred(sm, env_rank=i)
    res = DeferredResolver(AudioEngine(sr, device="cuda:0", rir_slots=slots), fast=True, prefetch_azimuths=False)
    res.native_miss_path = in_call
    return sims, res


sims_a, res_a = make(True, 40)            # 196 poses through 40 entries: evicts all the time
sims_b, res_b = make(False, 256)
bad = 0
for k in range(steps):
    n_act = int(rng.integers(1, n_env + 1))
    for i in range(n_env):
        if rng.random() < 0.5:
            p = (int(rng.integers(0, n_nodes)), int(rng.integers(0, n_nodes)), int(rng.choice([0, 90, 180, 270])), f"s{int(rng.integers(0, 3))}.wav")
            for sm in (sims_a[i], sims_b[i]):
                sm._receiver_position_index, sm._source_position_index, sm.azimuth_angle, sm._current_sound = p
                sm._episode_step_count += 1
    outs = []
    for sims, res in ((sims_a, res_a), (sims_b, res_b)):
        reqs = [pickle.loads(pickle.dumps(sm.get_current_spectrogram_observation(None))) for sm in sims[:n_act]]
        o = res.resolve(reqs, want_audiogoal=True)
        outs.append((o["audiogoal"].cpu().numpy(), o["spectrogram"].cpu().numpy()))
    if not (np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])):
        bad += 1
st = res_a.engine.store
ok_books = sorted(st._slot_of.values()) == sorted(np.flatnonzero(st._used).tolist()) and len(st._free) + len(st._slot_of) == st.slots
print(f"miss-path soak seed {seed}: {steps} steps, {bad} mismatching, library loaded {res_a.library_loaded} poses "
      f"(store misses {st.misses}, entries {st.slots}), books consistent: {ok_books}")
import shutil
shutil.rmtree(td, ignore_errors=True)
sys.exit(1 if bad or not ok_books else 0)

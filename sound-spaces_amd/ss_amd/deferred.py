"""Deferred mode: batched audio for MULTI-PROCESS vector envs.

The reference's default vector env is ``habitat.VectorEnv`` (ss_baselines/common/env_utils.py:91-107,
``USE_SYNC_VECENV`` False by default, av_nav/config/default.py:48): every env lives in a worker PROCESS, its sensors run
there, and the finished observations travel to the trainer through a pipe.  An audio sensor that computes in the worker
would mean N processes each launching batch-1 kernels on their own GPU context.  Deferred mode keeps the plugin API
(same sensor classes, same registry names, same uuids) but moves the arithmetic to where the batch is:

* WORKER side: ``attach_deferred(sim)`` makes ``sim.get_current_spectrogram_observation`` /
  ``get_current_audiogoal_observation`` return an ``AudioRequest`` — what ``_compute_audiogoal`` WOULD read right now
  (soundspaces/simulator.py:608-666; continuous_simulator.py:413-456): sound name, clip window, RIR file key or the
  live RIR itself, silence, distractor — a few hundred bytes (SS1.0) instead of a 128 KB waveform / 13.5 KB
  spectrogram.  The clip itself is shipped the first time a worker uses a sound.  ``_audio_index`` is advanced in the
  worker exactly where the reference advances it.
* TRAINER side: ``DeferredResolver.resolve(requests)`` turns the N requests of a vector step into N ``UnitRequest`` s
  (source registry, RIR store with the wav reader, live-RIR slots per env) and ONE launch; ``resolve_into(rollouts,
  observations)`` / ``batch_obs`` write the result straight into the rollout rows, as the in-process observers do.

Caches: the reference memoises per (source, receiver, azimuth) inside each simulator (simulator.py:678-701).  In
deferred mode every step is rendered (the cache-miss path); a cache hit in the reference returns the same array, so the
results are identical except for the documented multi-second quirk (a cached entry freezes the clip
window first seen at a pose)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .renderer import UnitRequest


@dataclass
class AudioRequest:
    """Picklable stand-in for an audio observation (travels through habitat.VectorEnv's pipe)."""
    env: int = 0                                  # worker / env rank: key of the env's live-RIR slots
    kind: str = "spectrogram"                     # which sensor asked: "spectrogram" | "audiogoal"
    silent: bool = False
    sound: Optional[str] = None
    clip: Optional[np.ndarray] = None             # only the first time this worker uses `sound`
    t0: int = 0
    rir_key: Optional[str] = None                 # SS1.0: path of the binaural RIR wav (simulator.py:615-616)
    dis_sound: Optional[str] = None
    dis_clip: Optional[np.ndarray] = None
    dis_rir_key: Optional[str] = None
    live_rir: Optional[np.ndarray] = None         # SS2.0 / habitat_sim audio sensor: the RIR itself, [L, 2] float32
    last_rir: Optional[np.ndarray] = None         # SS2.0 CROSSFADE: the previous step's RIR
    wrap: Optional[bool] = None
    last_wrap: Optional[bool] = None


class DeferredSimAudio:
    """Worker-side adapter for ``SoundSpacesSim`` (continuous=False) or ``ContinuousSoundSpacesSim``."""

    def __init__(self, sim, env_rank: int = 0, continuous: bool = False):
        self.sim, self.env, self.continuous = sim, env_rank, continuous
        self._sent = set()                        # sounds whose clip the trainer already has

    def _clip_once(self, name, clip):
        if name in self._sent:
            return None
        self._sent.add(name)
        return np.ascontiguousarray(clip, dtype=np.float32)

    def request(self, kind: str) -> AudioRequest:
        """One request per simulator state: the second sensor of a step (AudioGoalSensor + SpectrogramSensor are both
        configured in savi) gets the SAME request back, so ``_audio_index`` advances once per step, as it does in the
        reference where the second sensor hits the per-pose cache (simulator.py:678-701)."""
        sim = self.sim
        if self.continuous:
            key = (sim._episode_step_count, int(sim._current_sample_index), id(sim._prev_sim_obs))
        else:
            key = (sim._episode_step_count, sim._receiver_position_index, sim._source_position_index,
                   sim.azimuth_angle, sim._current_sound)
        if getattr(self, "_memo_key", None) == key:
            return self._memo
        self._memo_key, self._memo = key, self._request(kind)
        return self._memo

    def _request(self, kind: str) -> AudioRequest:
        sim = self.sim
        sr = int(sim.config.AUDIO.RIR_SAMPLING_RATE)
        if sim._episode_step_count > sim._duration:                                  # simulator.py:610 / cont. :415
            return AudioRequest(env=self.env, kind=kind, silent=True)
        name, clip = sim._current_sound, sim.current_source_sound
        req = AudioRequest(env=self.env, kind=kind, sound=name, clip=self._clip_once(name, clip))
        if self.continuous:
            rir = np.transpose(np.asarray(sim._prev_sim_obs["audio_sensor"], dtype=np.float32))     # cont. :419
            req.t0 = int(sim._current_sample_index)
            req.live_rir = np.ascontiguousarray(rir)
            req.wrap = req.t0 - rir.shape[0] >= 0                                    # cont. :433 vs :438-447
            last = getattr(sim, "_last_rir", None)
            if sim.config.AUDIO.CROSSFADE and last is not None:                      # cont. :422
                req.last_rir = np.ascontiguousarray(last, dtype=np.float32)
                req.last_wrap = req.t0 - req.last_rir.shape[0] >= 0
            return req
        if clip.shape[0] != sr:                                                      # simulator.py:634-635
            req.t0 = sim._audio_index * sr
            sim._audio_index = (sim._audio_index + 1) % sim._audio_length
        if sim.config.USE_RENDERED_OBSERVATIONS:
            req.rir_key = os.path.join(sim.binaural_rir_dir, str(sim.azimuth_angle),
                                       "{}_{}.wav".format(sim._receiver_position_index, sim._source_position_index))
        else:                                                                        # simulator.py:626
            req.live_rir = np.ascontiguousarray(np.transpose(np.array(sim._sim.get_sensor_observations()["audio_sensor"])),
                                                dtype=np.float32)
        if sim.config.AUDIO.HAS_DISTRACTOR_SOUND:                                    # simulator.py:649-664
            dn = sim._current_distractor_sound
            req.dis_sound = dn
            req.dis_clip = self._clip_once(dn, sim._source_sound_dict[dn])
            req.dis_rir_key = os.path.join(sim.binaural_rir_dir, str(sim.azimuth_angle),
                                           "{}_{}.wav".format(sim._receiver_position_index, sim._distractor_position_index))
        return req

    def get_current_audiogoal_observation(self):
        return self.request("audiogoal")

    def get_current_spectrogram_observation(self, audiogoal2spectrogram=None):
        return self.request("spectrogram")


def attach_deferred(sim, env_rank: int = 0, continuous: bool = False) -> DeferredSimAudio:
    """Worker side: the task sensors (the reference's or ss_amd's) keep calling ``sim.get_current_*_observation`` and now
    get an ``AudioRequest`` back, which habitat ships to the trainer as the sensor's 'observation'."""
    backend = DeferredSimAudio(sim, env_rank, continuous)
    sim.get_current_audiogoal_observation = backend.get_current_audiogoal_observation
    sim.get_current_spectrogram_observation = backend.get_current_spectrogram_observation
    sim._ss_hip_audio = backend
    return backend


class DeferredResolver:
    """Trainer side: N requests -> one launch.  ``engine`` = ``ss_amd.renderer.AudioEngine`` (an SS2.0 one for
    continuous simulators); ``rir_reader(path) -> [L, 2] array or None`` as in ``sim_audio.wav_rir_reader``."""

    def __init__(self, engine, rir_reader: Optional[Callable[[str], Optional[np.ndarray]]] = None):
        from .sim_audio import wav_rir_reader
        self.engine = engine
        self.rir_reader = rir_reader or wav_rir_reader
        self._clips: Dict[str, np.ndarray] = {}
        self._live: Dict[int, list] = {}          # env -> [held arrays, slots, turn] (see HipContinuousSimAudio)

    def _sound(self, name, clip) -> int:
        if clip is not None:
            self._clips[name] = clip
        if name not in self._clips:
            raise KeyError(f"deferred audio: the clip of sound {name!r} never arrived (worker restarted?)")
        return self.engine.source_id(name, self._clips[name])

    def _live_slot(self, env: int, rir: np.ndarray, avoid: int = -1) -> int:
        held, slots, turn = self._live.setdefault(env, [[None, None], [-1, -1], [0]])
        for k in (0, 1):
            h = held[k]
            if h is not None and (h is rir or (h.shape == rir.shape and np.array_equal(h, rir))):
                # through the store even on a content match: touches the LRU, marks the slot as used by this batch and
                # re-uploads the row if the store has meanwhile given the slot to another key
                slots[k] = self.engine.rir_slot(("live", env, k), lambda: rir, refresh=False)
                return slots[k]
        k = turn[0]
        if slots[k] == avoid and avoid >= 0:
            k ^= 1
        turn[0] = k ^ 1
        held[k] = rir
        slots[k] = self.engine.rir_slot(("live", env, k), lambda: rir, refresh=True)
        return slots[k]

    def units(self, requests: Sequence[AudioRequest]) -> List[UnitRequest]:
        if hasattr(self.engine, "begin_batch"):
            self.engine.begin_batch()
        out = []
        for q in requests:
            if q.silent:
                out.append(UnitRequest(silent=True))
                continue
            u = UnitRequest(sound=self._sound(q.sound, q.clip), t0=q.t0, wrap=q.wrap, last_wrap=q.last_wrap)
            if q.live_rir is not None:
                u.rir = self._live_slot(q.env, q.live_rir)
                if q.last_rir is not None:
                    u.last_rir = self._live_slot(q.env, q.last_rir, avoid=u.rir)
            else:
                u.rir = self.engine.rir_slot(q.rir_key, lambda key=q.rir_key: self.rir_reader(key))
            if q.dis_rir_key is not None:
                u.dis_sound = self._sound(q.dis_sound, q.dis_clip)
                u.dis_rir = self.engine.rir_slot(q.dis_rir_key, lambda key=q.dis_rir_key: self.rir_reader(key))
            out.append(u)
        return out

    def resolve(self, requests: Sequence[AudioRequest], want_audiogoal: bool = False, want_spectrogram: bool = True,
                spectrogram_out=None, audiogoal_out=None):
        """-> {"spectrogram": [N,65,T4,2], ("audiogoal": [N,2,sr])} device tensors, one launch for all envs."""
        return self.engine.observe(self.units(requests), want_audiogoal=want_audiogoal or audiogoal_out is not None,
                                   want_spectrogram=want_spectrogram, spectrogram_out=spectrogram_out,
                                   audiogoal_out=audiogoal_out)

    def resolve_observations(self, observations: Sequence[dict], rollouts=None):
        """Replacement for the audio half of ``batch_obs`` (ss_baselines/common/utils.py:126-153): `observations` is
        the list of per-env dicts the vector env returned; entries under 'spectrogram' / 'audiogoal' that are
        ``AudioRequest`` s are rendered in one launch (into the rollout rows of the next ``insert()`` when `rollouts` is
        given) and REPLACED, per env, by their device tensors (views of the batch), so the dicts can go on to
        ``batch_obs`` unchanged.  Returns the batched tensors {uuid: [N, ...]}."""
        keys = [k for k in ("spectrogram", "audiogoal") if observations and isinstance(observations[0].get(k), AudioRequest)]
        if not keys:
            return {}
        reqs = [obs[keys[0]] for obs in observations]
        slots = rollouts.next_observation_slots([k for k in keys if k in rollouts.observations]) if rollouts is not None else {}
        out = self.resolve(reqs, want_audiogoal="audiogoal" in keys, want_spectrogram="spectrogram" in keys,
                           spectrogram_out=slots.get("spectrogram"), audiogoal_out=slots.get("audiogoal"))
        for k in keys:
            for i, obs in enumerate(observations):
                obs[k] = out[k][i]
        return {k: out[k] for k in keys}

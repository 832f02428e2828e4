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
import zlib
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .renderer import UnitRequest


import functools
import operator

_LIVE_GET = operator.attrgetter("live_rir", "last_rir")


def _n_none(items) -> int:
    """how many entries ARE None (identity: the entries may be arrays, whose == is element-wise)"""
    return sum(map(_IS_NONE, items))


_IS_NONE = functools.partial(operator.is_, None)


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
    # SS2.0: sequence numbers of the worker's live RIRs (> 0).  `_last_rir` is the RIR the worker sent as `live_rir` one
    # request earlier (continuous_simulator.py:384) - the worker checks that (its own compare, in its own process) and says
    # so with `last_seq = live_seq - 1`: the trainer finds the row it already holds by number instead of comparing 72 KB
    # against each held row per env and step
    live_seq: int = 0
    last_seq: int = 0
    wrap: Optional[bool] = None
    last_wrap: Optional[bool] = None
    # SS1.0 requests against RIR files: the same facts as REC_N packed int64 words (bytes), so that the trainer turns the
    # N requests of a vector step into unit COLUMNS with a dozen numpy operations instead of a Python walk (see
    # DeferredResolver._columns).  Names travel as CRC-32 keys; the strings above stay for first-use registration.
    rec: Optional[bytes] = None
    # numbered live RIRs (SS2.0 workers): the scalar facts of the request as LREC_N packed int64 words, so that the trainer
    # reads a step's N requests with one bytes.join + frombuffer and touches the request objects only for their arrays
    lrec: Optional[bytes] = None
    # pose_cache mode (attach_deferred(..., pose_cache=True)): the reference's per-pose memo (simulator.py:678-701) kept by
    # the worker's own simulator dicts; a hit asks the trainer for the row it stored when this pose was first rendered
    cache_key: Optional[tuple] = None             # (source, receiver, azimuth)
    cache_hit: bool = False
    cache_epoch: int = 0                          # bumps when the simulator dropped its caches (scene / sound change, :395-397)


# layout of AudioRequest.rec (int64 words)
REC_SILENT, REC_SOUND, REC_T0, REC_TABLE, REC_RECV, REC_SRC, REC_DIS_SOUND, REC_DIS_SRC, REC_ENV, REC_RESERVED = range(10)
REC_N = 10                                    # = SS_REQ_WORDS of include/ss_hip.h
_SILENT_REC = np.zeros((REC_N,), np.int64)
_SILENT_REC[REC_SILENT] = 1
_SILENT_REC[REC_DIS_SOUND] = -1


# layout of AudioRequest.lrec (int64 words); LREC_LAST_WRAP = -1: no previous RIR (first step of an episode, CROSSFADE off)
LREC_SILENT, LREC_SOUND, LREC_T0, LREC_ENV, LREC_SEQ, LREC_LAST_SEQ, LREC_WRAP, LREC_LAST_WRAP = range(8)
LREC_N = 8


def _silent_rec(env: int) -> bytes:
    rec = _SILENT_REC.copy()
    rec[REC_ENV] = env
    return rec.tobytes()


def name_key(name: str) -> int:
    """CRC-32 of a sound name / RIR directory: the same integer in every process without a registry round trip."""
    return zlib.crc32(name.encode("utf-8"))


class DeferredSimAudio:
    """Worker-side adapter for ``SoundSpacesSim`` (continuous=False) or ``ContinuousSoundSpacesSim``."""

    def __init__(self, sim, env_rank: int = 0, continuous: bool = False, pose_cache: bool = False):
        self.sim, self.env, self.continuous = sim, env_rank, continuous
        # reference-exact mode for multi-second sounds (see FastVectorAudioObserver(pose_cache=)): SoundSpaces 1.0 without
        # a distractor only - the reference bypasses its caches otherwise (:679-681), SS2.0 has none (cont. :458-462)
        self.pose_cache = bool(pose_cache) and not continuous
        # cache_epoch = a number no earlier incarnation of this worker can have used (a restarted worker that
        # started from 0 again would be served the rows its predecessor stored under the same epoch)
        self._epoch, self._cache_obj = int.from_bytes(os.urandom(4), "little") << 20, None
        self._sent = set()                        # sounds whose clip the trainer already has
        self._keys: Dict[str, int] = {}           # name -> CRC-32 (sounds, "<rir dir>/<azimuth>" tables)

    def _key(self, name: str) -> int:
        k = self._keys.get(name)
        if k is None:
            k = self._keys[name] = name_key(name)
        return k

    def _clip_once(self, name, clip):
        if name in self._sent:
            return None
        self._sent.add(name)
        return np.ascontiguousarray(clip, dtype=np.float32)

    def request(self, kind: str) -> AudioRequest:
        """One request per simulator state: the second sensor of a step (AudioGoalSensor + SpectrogramSensor both
        configured) gets the SAME request back, so ``_audio_index`` advances once per step, as it does in the reference where
        the second sensor hits the per-pose cache (simulator.py:682-686, 694-698).  With HAS_DISTRACTOR_SOUND the reference does
        NOT cache (:679-681, :691-693): every sensor read computes again and advances ``_audio_index`` again - so does this
        adapter (a new request per read; ``DeferredResolver.resolve_observations`` renders the two sensors' requests
        separately when they name different clip windows)."""
        sim = self.sim
        if self.continuous:
            key = (sim._episode_step_count, int(sim._current_sample_index), id(sim._prev_sim_obs))
        else:
            key = (sim._episode_step_count, sim._receiver_position_index, sim._source_position_index,
                   sim.azimuth_angle, sim._current_sound)
        uncached = not self.continuous and bool(sim.config.AUDIO.HAS_DISTRACTOR_SOUND)
        if not uncached and getattr(self, "_memo_key", None) == key:
            return self._memo
        if self.pose_cache and not sim.config.AUDIO.HAS_DISTRACTOR_SOUND:
            cache = sim._spectrogram_cache                    # the simulator's own dict: reconfigure() replaces it (:395-397)
            if cache is not self._cache_obj or (not cache and self._cache_obj is not None and self._had_entries):
                self._epoch += 1
                self._cache_obj = cache
            self._had_entries = bool(cache)
            pose = (sim._source_position_index, sim._receiver_position_index, sim.azimuth_angle)      # :683
            if pose in cache:                                 # hit: nothing is computed, _audio_index stays (:634-635)
                req = AudioRequest(env=self.env, kind=kind, cache_key=pose, cache_hit=True, cache_epoch=self._epoch)
            else:
                req = self._request(kind)
                req.cache_key, req.cache_epoch = pose, self._epoch
                cache[pose] = req                             # marks the pose as rendered (the value is never read)
                self._had_entries = True
            self._memo_key, self._memo = key, req
            return req
        self._memo_key, self._memo = key, self._request(kind)
        return self._memo

    def _request(self, kind: str) -> AudioRequest:
        sim = self.sim
        sr = int(sim.config.AUDIO.RIR_SAMPLING_RATE)
        if sim._episode_step_count > sim._duration:                                  # simulator.py:610 / cont. :415
            rec = _SILENT_REC.copy()
            rec[REC_ENV] = self.env
            lrec = None
            if self.continuous:
                lrec = np.zeros((LREC_N,), np.int64)
                lrec[LREC_SILENT], lrec[LREC_ENV], lrec[LREC_LAST_WRAP] = 1, self.env, -1
                lrec = lrec.tobytes()
            return AudioRequest(env=self.env, kind=kind, silent=True, rec=rec.tobytes(), lrec=lrec)
        name, clip = sim._current_sound, sim.current_source_sound
        req = AudioRequest(env=self.env, kind=kind, sound=name, clip=self._clip_once(name, clip))
        if self.continuous:
            rir = np.transpose(np.asarray(sim._prev_sim_obs["audio_sensor"], dtype=np.float32))     # cont. :419
            req.t0 = int(sim._current_sample_index)
            req.live_rir = np.ascontiguousarray(rir)
            req.wrap = req.t0 - rir.shape[0] >= 0                                    # cont. :433 vs :438-447
            self._live_seq = getattr(self, "_live_seq", 0) + 1                       # (one _request per simulator state: request())
            req.live_seq = self._live_seq
            last = getattr(sim, "_last_rir", None)
            if sim.config.AUDIO.CROSSFADE and last is not None:                      # cont. :422
                last = np.ascontiguousarray(last, dtype=np.float32)
                req.last_wrap = req.t0 - last.shape[0] >= 0
                prev = getattr(self, "_live_prev", None)
                from .sim_audio import same_rir
                req.last_rir = last
                if prev is not None and same_rir(prev, last):                        # the array of the previous request: the
                    req.last_seq = self._live_seq - 1                                # trainer holds it under that number
            self._live_prev = req.live_rir
            lrec = np.zeros((LREC_N,), np.int64)
            lrec[LREC_SOUND], lrec[LREC_T0], lrec[LREC_ENV] = self._key(name), req.t0, self.env
            lrec[LREC_SEQ], lrec[LREC_LAST_SEQ], lrec[LREC_WRAP] = req.live_seq, req.last_seq, bool(req.wrap)
            lrec[LREC_LAST_WRAP] = -1 if req.last_rir is None else int(bool(req.last_wrap))
            req.lrec = lrec.tobytes()
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
        if req.rir_key is not None:                       # RIR files: (directory/azimuth, receiver, source) are integers
            rec = np.zeros((REC_N,), np.int64)
            rec[REC_SOUND] = self._key(name)
            rec[REC_T0] = req.t0
            rec[REC_TABLE] = self._key(os.path.join(sim.binaural_rir_dir, str(sim.azimuth_angle)))
            rec[REC_RECV] = sim._receiver_position_index
            rec[REC_SRC] = sim._source_position_index
            rec[REC_DIS_SOUND] = self._key(req.dis_sound) if req.dis_sound is not None else -1
            rec[REC_DIS_SRC] = sim._distractor_position_index if req.dis_sound is not None else 0
            rec[REC_ENV] = self.env
            req.rec = rec.tobytes()
        return req

    def get_current_audiogoal_observation(self):
        return self.request("audiogoal")

    def get_current_spectrogram_observation(self, audiogoal2spectrogram=None):
        return self.request("spectrogram")


def attach_deferred(sim, env_rank: int = 0, continuous: bool = False, pose_cache: bool = False) -> DeferredSimAudio:
    """Worker side: the task sensors (the reference's or ss_amd's) keep calling ``sim.get_current_*_observation`` and now
    get an ``AudioRequest`` back, which habitat ships to the trainer as the sensor's 'observation'."""
    backend = DeferredSimAudio(sim, env_rank, continuous, pose_cache)
    sim.get_current_audiogoal_observation = backend.get_current_audiogoal_observation
    sim.get_current_spectrogram_observation = backend.get_current_spectrogram_observation
    sim._ss_hip_audio = backend
    return backend


class DeferredResolver:
    """Trainer side: N requests -> one launch.  ``engine`` = ``ss_amd.renderer.AudioEngine`` (an SS2.0 one for
    continuous simulators); ``rir_reader(path) -> [L, 2] array or None`` as in ``sim_audio.wav_rir_reader``."""

    PREFETCH_MAX_POSES = 4                      # new poses per step up to which their sibling azimuths are loaded along

    def __init__(self, engine, rir_reader: Optional[Callable[[str], Optional[np.ndarray]]] = None,
                 fast: Optional[bool] = None, prefetch_azimuths: bool = True):
        """fast: None = use the column path when the engine has one and every request of the step carries `rec`;
        False = always walk the requests (per-unit Python planner); True = require the column path.
        prefetch_azimuths (column path): a pose that is not resident is loaded together with the OTHER azimuths of its
        (receiver, source) pair - `<binaural_rir_dir>/{0,90,180,270}/<recv>_<src>.wav`, simulator.py:615-616 - in the same
        read + upload: two of the agent's three actions are turns (`TURN_LEFT` / `TURN_RIGHT` keep the node), so the step after
        a miss is usually a rotation of the same pair, and a miss STEP costs far more than a file (fixed ~0.2 ms against
        ~20 us per file: scripts/bench_loader.py).  Only for steps with at most PREFETCH_MAX_POSES new poses - the reference's
        5-10 envs per GPU, where the prefetch turns most steps into plain hits; a 128-env step has some env on a new pose
        nearly every time, pays the fixed cost anyway and would only read the files earlier (measured: -7 %).  Files that do
        not exist are simply not prefetched."""
        from .sim_audio import wav_rir_reader
        self.engine = engine
        self.prefetch_azimuths = bool(prefetch_azimuths)
        self._siblings: Dict[int, list] = {}      # table id -> ids of the other azimuth tables of its RIR directory
        self.prefetched = 0
        self.rir_reader = rir_reader or wav_rir_reader
        self._clips: Dict[str, np.ndarray] = {}
        self._live: Dict[int, list] = {}          # env -> [held arrays, slots, turn] (see HipContinuousSimAudio)
        # numbered live RIRs (SS2.0 workers): env e keeps two bank rows, RIR number q lives in row (q & 1) - the current RIR
        # and the previous step's one (CROSSFADE) never share a row, and a step of N envs is N row writes at known places
        self._lv_slots = np.full((0, 2), -1, np.int64)     # [env, parity] -> store slot (-1: not allocated)
        self._lv_seqs = np.zeros((0, 2), np.int64)         # [env, parity] -> number of the RIR the row holds (0: none)
        self._sid: Dict[str, int] = {}                      # sound name -> source id (live path)
        self.live_steps = 0
        # column path (engines that own a C++ context: ss_amd.renderer.AudioEngine on a GPU).  CRC keys -> ids through
        # sorted arrays (np.searchsorted), RIR slots through dense (table, receiver, source) tables
        store = getattr(engine, "store", None)
        self.columns_ok = (fast is not False and hasattr(engine, "observe_columns") and store is not None
                           and getattr(store, "group", 0) == 1 and hasattr(store, "touch_slots"))
        if fast is True and not self.columns_ok:
            raise ValueError("DeferredResolver(fast=True): the engine has no column path (AudioEngine, rir_group=1)")
        self._sound_keys = np.zeros((0,), np.int64)
        self._sound_ids = np.zeros((0,), np.int64)
        self._key_names: Dict[int, str] = {}
        self._table_keys = np.zeros((0,), np.int64)
        self._table_ids = np.zeros((0,), np.int64)
        self._table_dirs: List[str] = []
        self._native_reader = None                          # (rir_reader, stock wav reader?, lenient?), decided on first use
        self.column_steps = self.walk_steps = 0
        # resident RIR files: sorted composite keys (table << 40 | receiver << 20 | source) -> store slot
        self._pair_buf = None                               # (keys, slots) buffers with spare capacity behind _pk / _ps (_request_tables)
        self._loader = None                                 # RirStore.miss_loader dict (ss_ctx_observe_requests_load)
        self._loader_ok = None
        self.native_miss_path = True                        # False: pose misses always take report -> load_files -> call again
        self.library_loaded = 0                             # poses the library loaded inside the step's call
        self._pk = np.zeros((0,), np.int64)                 # resident (table, receiver, source) keys, sorted: _pair_keys
        self._ps = np.zeros((0,), np.int64)                 # ... and their store slots: _pair_slots
        self._evq: List[int] = []                           # keys the store has evicted since the arrays were last read
        self._tables = None                       # the arrays above as the C struct of ss_ctx_observe_requests (rebuilt on change)
        # pose_cache workers: env -> [epoch, {pose: pool row}], pools of cached output rows on the device
        self._pose_maps: Dict[int, list] = {}
        self._pose_pool: Dict[str, object] = {}
        self._pose_free: List[int] = []
        self._pose_cap = 0
        self.native_steps = self.miss_steps = 0
        if self.columns_ok:                       # every resolver over this store hears about evictions (their pair tables
            if hasattr(store, "add_evict_hook"):  # name store slots); held weakly: a dropped resolver drops out
                store.add_evict_hook(self._evicted)
            else:
                store.on_evict = self._evicted

    def _sound(self, name, clip) -> int:
        if clip is not None:
            self._clips[name] = clip
        if name not in self._clips:
            raise KeyError(f"deferred audio: the clip of sound {name!r} never arrived (worker restarted?)")
        return self.engine.source_id(name, self._clips[name])

    def _lv_rows(self, env_max: int) -> None:
        if env_max >= self._lv_slots.shape[0]:
            grow = max(env_max + 1, 2 * self._lv_slots.shape[0], 64)
            sl, sq = np.full((grow, 2), -1, np.int64), np.zeros((grow, 2), np.int64)
            sl[:self._lv_slots.shape[0]], sq[:self._lv_seqs.shape[0]] = self._lv_slots, self._lv_seqs
            self._lv_slots, self._lv_seqs = sl, sq

    def _live_slot_seq(self, env: int, seq: int, rir: Optional[np.ndarray], par: Optional[int] = None) -> int:
        """Bank slot of env's live RIR number `seq` (workers that number their RIRs): row (seq & 1) of the env's two, no
        content compares.  rir = None: the row must still be held (it was this env's `live_rir` one request ago).  `par`:
        the row to use for an un-numbered companion (a `last_rir` sent in full: the row the current RIR does not use)."""
        self._lv_rows(env)
        k = (seq & 1) if par is None else par
        held = self._lv_seqs[env, k] == seq and self._lv_slots[env, k] >= 0
        if not held and rir is None:
            raise KeyError(f"deferred audio: env {env}'s live RIR {seq} is no longer in the store (rir_slots < 2 per env?)")
        def gone():
            raise KeyError(f"deferred audio: env {env}'s live RIR {seq} is no longer in the store (rir_slots < 2 per env?)")
        slot = self.engine.rir_slot(("live", env, k), (lambda: rir) if rir is not None else gone, refresh=not held)
        self._lv_slots[env, k], self._lv_seqs[env, k] = slot, seq
        return slot

    def _live_slot(self, env: int, rir: np.ndarray, avoid: int = -1) -> int:
        from .sim_audio import same_rir
        held, slots, turn = self._live.setdefault(env, [[None, None], [-1, -1], [0]])
        for k in (0, 1):
            h = held[k]
            if h is not None and same_rir(h, rir):
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
            if q.live_rir is not None and q.live_seq > 0:                 # numbered live RIRs (SS2.0 workers of this version)
                u.rir = self._live_slot_seq(q.env, q.live_seq, q.live_rir)
                if q.last_seq > 0:
                    u.last_rir = self._live_slot_seq(q.env, q.last_seq, q.last_rir)
                elif q.last_rir is not None:                                  # not the previous request's array: sent in full
                    u.last_rir = self._live_slot_seq(q.env, -q.live_seq, q.last_rir, par=(q.live_seq & 1) ^ 1)
            elif q.live_rir is not None:
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

    # ---- column path ---------------------------------------------------------------------------------------
    @staticmethod
    def _lookup(keys: np.ndarray, ids: np.ndarray, q: np.ndarray) -> np.ndarray:
        """ids of the CRC keys q (sorted `keys`), -1 where unknown"""
        if keys.shape[0] == 0:
            return np.full(q.shape, -1, np.int64)
        pos = np.minimum(np.searchsorted(keys, q), keys.shape[0] - 1)
        return np.where(keys[pos] == q, ids[pos], -1)

    def _learn_sound(self, name: str, clip) -> None:
        key = name_key(name)
        if self._key_names.setdefault(key, name) != name:
            raise KeyError(f"deferred audio: sounds {name!r} and {self._key_names[key]!r} share a CRC-32 key")
        sid = self._sound(name, clip)
        order = np.argsort(np.append(self._sound_keys, key), kind="stable")
        self._sound_keys = np.append(self._sound_keys, key)[order]
        self._sound_ids = np.append(self._sound_ids, sid)[order]
        self._tables = None

    def _learn_table(self, rir_key: str) -> None:
        d = os.path.dirname(rir_key)                        # <binaural_rir_dir>/<azimuth>
        key = name_key(d)
        if self._key_names.setdefault(key, d) != d:
            raise KeyError(f"deferred audio: {d!r} and {self._key_names[key]!r} share a CRC-32 key")
        tid = len(self._table_dirs)
        self._table_dirs.append(d)
        order = np.argsort(np.append(self._table_keys, key), kind="stable")
        self._table_keys = np.append(self._table_keys, key)[order]
        self._table_ids = np.append(self._table_ids, tid)[order]
        self._tables = None

    @staticmethod
    def _pair_key(table, recv, src):
        return (table << 40) | (recv << 20) | src

    def _evicted(self, key, slot) -> None:
        if isinstance(key, tuple) and len(key) == 3 and key[0] == "live":          # a live row lost its slot
            if key[1] < self._lv_slots.shape[0]:
                self._lv_slots[key[1], key[2]], self._lv_seqs[key[1], key[2]] = -1, 0
            return
        if isinstance(key, tuple) and len(key) == 2 and key[0] == "ix":
            # queued: a miss step against a full store evicts one pose per new pose, and one np.delete per array and victim
            # was most of what such a step cost beyond a step into free slots (profiles/r5/NOTES.md section 2)
            self._evq.append(key[1])
            self._tables = None                             # (the C tables point at arrays that still hold the key)

    def _drop_evicted(self) -> None:
        ev = np.asarray(self._evq, np.int64)
        self._evq = []
        n = self._pk.shape[0]
        if n == 0:
            return
        pos = np.minimum(np.searchsorted(self._pk, ev), n - 1)
        pos = pos[self._pk[pos] == ev]
        if pos.shape[0]:
            keep = np.ones((n,), bool)
            keep[pos] = False
            self._pk, self._ps = self._pk[keep], self._ps[keep]

    @property
    def _pair_keys(self) -> np.ndarray:
        if self._evq:
            self._drop_evicted()
        return self._pk

    @_pair_keys.setter
    def _pair_keys(self, v) -> None:
        self._pk = v

    @property
    def _pair_slots(self) -> np.ndarray:
        if self._evq:
            self._drop_evicted()
        return self._ps

    @_pair_slots.setter
    def _pair_slots(self, v) -> None:
        self._ps = v

    def _load_pairs(self, pair_keys: np.ndarray, which: np.ndarray, reload: bool = False) -> None:
        """RIR files of the composite keys at positions `which` -> store slots -> the resident-pair arrays (the
        first-visit side of the column path, simulator.py:615-618).  With the stock wav reader all new poses of the step are
        read by the library's native reader in one call (``RirStore.load_files``: plain threads, one pinned block, one H2D
        copy) and enter the sorted arrays with ONE merge; a custom reader keeps the file-by-file walk."""
        store = self.engine.store
        ks = np.unique(np.asarray(pair_keys)[which].astype(np.int64))
        if ks.shape[0] == 0:
            return
        if self.prefetch_azimuths and not reload and ks.shape[0] <= self.PREFETCH_MAX_POSES:
            ks = self._with_sibling_azimuths(ks)
        kl = ks.tolist()
        dirs = self._table_dirs
        paths = [os.path.join(dirs[k >> 40], f"{(k >> 20) & 0xFFFFF}_{k & 0xFFFFF}.wav") for k in kl]
        if self._native_reader is None or self._native_reader[0] is not self.rir_reader:
            from .renderer import _native_wav
            import functools
            self._native_reader = (self.rir_reader, bool(_native_wav(self.rir_reader)),
                                   isinstance(self.rir_reader, functools.partial) and bool(self.rir_reader.keywords.get("lenient")))
        if self._native_reader[1] and hasattr(store, "load_files") and store.group == 1:
            slots = store.load_files([("ix", k) for k in kl], paths, reader=self.rir_reader, missing_ok=self._native_reader[2],
                                     new_batch=False)
        else:
            slots = [store.slot(("ix", k), lambda path=path: self.rir_reader(path)) for k, path in zip(kl, paths)]
        slots = np.asarray(slots, np.int64)
        # loading may have evicted resident pairs (the hook removed them from the arrays): merge against what is there NOW
        n_old = self._pair_keys.shape[0]
        pos = np.searchsorted(self._pair_keys, ks)
        have = (self._pair_keys[np.minimum(pos, n_old - 1)] == ks) if n_old else np.zeros(ks.shape, bool)
        if have.any():
            self._pair_slots[pos[have]] = slots[have]
        if not have.all():                                  # ONE merge of the new keys into both sorted arrays
            new = ~have
            at = pos[new] + np.arange(int(new.sum()))
            keep = np.ones((n_old + at.shape[0],), bool)
            keep[at] = False
            keys2, slots2 = np.empty(keep.shape, np.int64), np.empty(keep.shape, np.int64)
            keys2[at], slots2[at] = ks[new], slots[new]
            keys2[keep], slots2[keep] = self._pair_keys, self._pair_slots
            self._pair_keys, self._pair_slots = keys2, slots2
        self._tables = None

    def preload_scene(self, binaural_rir_dir: str, azimuths=(0, 90, 180, 270), limit: Optional[int] = None) -> int:
        """Bulk load of one scene's RIR files, `<binaural_rir_dir>/<azimuth>/<receiver>_<source>.wav` (soundspaces/README.md:
        38-42, simulator.py:615-616), into the HBM store UNDER THIS RESOLVER'S KEYS, so that no step of an episode in the scene
        goes through the miss path (``renderer.load_scene_rirs`` keys the rows by file path: that serves the eager adapter).
        At most as many poses as the store has free entries (nothing resident is evicted for a pre-load), `limit` if given.
        Returns the number of poses loaded."""
        store = self.engine.store
        ks = []
        for az in azimuths:
            d = os.path.join(binaural_rir_dir, str(az))
            if not os.path.isdir(d):
                continue
            if name_key(d) not in self._key_names:
                self._learn_table(os.path.join(d, "x.wav"))
            elif self._key_names[name_key(d)] != d:
                raise KeyError(f"deferred audio: {d!r} and {self._key_names[name_key(d)]!r} share a CRC-32 key")
            t = self._table_dirs.index(d)
            for name in os.listdir(d):
                r, sep, s_ = name[:-4].partition("_")
                if name.endswith(".wav") and sep and r.isdigit() and s_.isdigit() and int(r) < (1 << 20) and int(s_) < (1 << 20):
                    ks.append((t << 40) | (int(r) << 20) | int(s_))
        ks = np.unique(np.asarray(ks, np.int64))
        ks = ks[self._lookup(self._pair_keys, self._pair_slots, ks) < 0] if ks.shape[0] else ks
        room = len(getattr(store, "_free", ())) if hasattr(store, "_free") else ks.shape[0]
        n = min(ks.shape[0], room, ks.shape[0] if limit is None else int(limit))
        for lo in range(0, n, 256):
            store.begin_batch()
            self._load_pairs(ks, np.arange(lo, min(lo + 256, n)), reload=True)      # (reload: no sibling prefetch on top)
        return n

    def _sibling_tables(self, t: int) -> list:
        """ids of the other azimuth directories next to table t's (`<binaural_rir_dir>/<azimuth>`), registered on first need"""
        sib = self._siblings.get(t)
        if sib is None:
            d = self._table_dirs[t]
            parent, az = os.path.split(d)
            sib = []
            if az in ("0", "90", "180", "270"):
                for other in ("0", "90", "180", "270"):
                    od = os.path.join(parent, other)
                    if other == az or not os.path.isdir(od):
                        continue
                    key = name_key(od)
                    if key not in self._key_names:
                        self._learn_table(os.path.join(od, "x.wav"))
                    elif self._key_names[key] != od:
                        continue
                    sib.append(self._table_dirs.index(od))
            self._siblings[t] = sib
        return sib

    def _with_sibling_azimuths(self, ks: np.ndarray) -> np.ndarray:
        """ks + the same (receiver, source) under the other azimuth directories, where that file exists and is not resident"""
        extra = []
        for k in ks.tolist():
            t, rest = k >> 40, k & ((1 << 40) - 1)
            for o in self._sibling_tables(t):
                k2 = (o << 40) | rest
                pos = int(np.searchsorted(self._pair_keys, k2))
                if pos < self._pair_keys.shape[0] and self._pair_keys[pos] == k2:
                    continue
                if os.path.exists(os.path.join(self._table_dirs[o], "{}_{}.wav".format((rest >> 20) & 0xFFFFF, rest & 0xFFFFF))):
                    extra.append(k2)
        store = self.engine.store
        room = len(getattr(store, "_free", ())) * getattr(store, "group", 1) - int(ks.shape[0])
        if not extra or room < len(extra):                  # best effort: never evict resident poses for a guess
            return ks
        self.prefetched += len(extra)
        return np.unique(np.concatenate([ks, np.asarray(extra, np.int64)]))

    def _live_columns(self, requests: Sequence[AudioRequest]):
        """A step whose requests carry NUMBERED live RIRs (SoundSpaces 2.0 workers: every env a new RIR every step,
        continuous_simulator.py:370-392, 413-426) -> unit columns
        for ``engine.observe_columns``, or None when the step needs the per-request walk (RIR files among the requests, a
        distractor, un-numbered RIRs).  Per request only one ``attrgetter`` call; row (seq & 1) of env e holds RIR number
        seq, so the N new RIRs of the step go to N known rows through ONE gathered upload (``RirStore.upload_rows``) and the
        previous step's RIR (CROSSFADE, :422-424) is found in the env's other row without comparing contents."""
        n = len(requests)
        try:
            buf = b"".join([q.lrec for q in requests])
        except TypeError:                                    # a request without the record (RIR files, un-numbered RIRs)
            return None
        recs = np.frombuffer(buf, np.int64).reshape(n, LREC_N)
        live = recs[:, LREC_SILENT] == 0
        n_silent = n - int(np.count_nonzero(live))
        if n_silent == n:
            return None
        seq, lseq, env, t0 = recs[:, LREC_SEQ], recs[:, LREC_LAST_SEQ], recs[:, LREC_ENV], recs[:, LREC_T0]
        rir, last = zip(*map(_LIVE_GET, requests))           # the only per-request attribute reads: the arrays themselves
        store = self.engine.store
        # sounds by CRC key (a clip travels with the first request that names it)
        sids = self._lookup(self._sound_keys, self._sound_ids, recs[:, LREC_SOUND])
        if (sids[live] < 0).any():
            for i in np.flatnonzero(live & (sids < 0)):
                q = requests[i]
                if name_key(q.sound) not in self._key_names:
                    self._learn_sound(q.sound, q.clip)
            sids = self._lookup(self._sound_keys, self._sound_ids, recs[:, LREC_SOUND])
            if (sids[live] < 0).any():
                raise KeyError("deferred audio: a live request names a sound that could not be registered")
        store.begin_batch()
        self._lv_rows(int(env.max()))
        par = seq & 1
        li = np.flatnonzero(live) if n_silent else np.arange(n)
        el, pl = env[li], par[li]
        unalloc = (self._lv_slots[el, 0] < 0) | (self._lv_slots[el, 1] < 0)
        if unalloc.any():
            #  this step's rows that already exist are marked in use BEFORE rows are allocated for the new envs: with a
            # full store the allocation may evict, and a victim among this step's own rows would reach the scatter as slot -1
            have = self._lv_slots[el[~unalloc]].ravel()
            if have.size:
                store.touch_slots(have)
            for e in np.unique(el[unalloc]):                     # an env's first step: its two rows
                for k in (0, 1):
                    if self._lv_slots[e, k] < 0:
                        self._lv_slots[e, k] = store.slot(("live", int(e), k), lambda: None, refresh=True)
                        self._lv_seqs[e, k] = 0
        cur = self._lv_slots[el, pl]
        if (cur < 0).any() or (self._lv_slots[el, pl ^ 1] < 0).any():
            raise KeyError("deferred audio: the RIR store is too small for this step's live rows (two per env); "
                           "an allocation evicted a row the step itself needs")
        new_cur = np.flatnonzero(self._lv_seqs[el, pl] != seq[li])
        up_slots = cur[new_cur].tolist()
        up_rows = [rir[j] for j in li[new_cur].tolist()]
        self._lv_seqs[el, pl] = seq[li]
        # the previous step's RIR: by number in the env's other row; an array that is NOT the previous request's (the worker
        # says so with last_seq = 0) is uploaded there under a number nothing else uses
        has_last = live & (recs[:, LREC_LAST_WRAP] >= 0)
        last_slot = None
        if has_last.any():
            hi = np.flatnonzero(has_last)
            eh, ph = env[hi], par[hi] ^ 1
            want = np.where(lseq[hi] > 0, lseq[hi], -seq[hi])
            stale = np.flatnonzero(self._lv_seqs[eh, ph] != want)
            last_slot = np.full((n,), -1, np.int64)
            last_slot[hi] = self._lv_slots[eh, ph]
            for j in hi[stale].tolist():
                up_slots.append(int(last_slot[j]))
                up_rows.append(np.ascontiguousarray(last[j], dtype=np.float32))
            self._lv_seqs[eh, ph] = want
            store.touch_slots(np.concatenate([cur, last_slot[hi]]))
        else:
            store.touch_slots(cur)
        if up_slots:
            store.upload_rows(up_slots, up_rows)
        if n_silent:
            rir_col = np.full((n,), -1, np.int64)
            rir_col[li] = cur
        else:
            rir_col = cur
        cols = dict(sound=np.where(live, sids, 0), t0=np.where(live, t0, 0), rir=rir_col, wrap=recs[:, LREC_WRAP].astype(np.uint8))
        if last_slot is not None:
            cols["last_rir"] = last_slot
            cols["last_wrap"] = np.maximum(recs[:, LREC_LAST_WRAP], 0).astype(np.uint8)
        return cols

    def _serve_misses(self, buf: bytes, n: int, requests, miss_idx: np.ndarray) -> bool:
        """The requests `miss_idx` of a step could not be resolved by ``ss_ctx_observe_requests``: register what is new among
        THEM (first use of a sound / RIR directory: needs the request objects) and load the RIR files of their poses that are
        not resident (or are clipped rows that are now needed whole).  Returns False when nothing could be done about them
        (the caller then takes the numpy path, which raises the precise error)."""
        recs = np.frombuffer(buf, np.int64).reshape(n, REC_N)[miss_idx]
        changed = False
        sound = self._lookup(self._sound_keys, self._sound_ids, recs[:, REC_SOUND])
        table = self._lookup(self._table_keys, self._table_ids, recs[:, REC_TABLE])
        has_dis = recs[:, REC_DIS_SOUND] >= 0
        dsound = self._lookup(self._sound_keys, self._sound_ids, recs[:, REC_DIS_SOUND]) if has_dis.any() else None
        new = (sound < 0) | (table < 0)
        if dsound is not None:
            new |= has_dis & (dsound < 0)
        if new.any():
            if requests is None:
                return False
            for j in np.flatnonzero(new):
                q = requests[int(miss_idx[j])]
                if sound[j] < 0 and name_key(q.sound) not in self._key_names:
                    self._learn_sound(q.sound, q.clip)
                    changed = True
                if table[j] < 0 and name_key(os.path.dirname(q.rir_key)) not in self._key_names:
                    self._learn_table(q.rir_key)
                    changed = True
                if dsound is not None and has_dis[j] and dsound[j] < 0 and name_key(q.dis_sound) not in self._key_names:
                    self._learn_sound(q.dis_sound, q.dis_clip)
                    changed = True
            table = self._lookup(self._table_keys, self._table_ids, recs[:, REC_TABLE])
            if (table < 0).any():
                return False
        store = self.engine.store
        pk = self._pair_key(table, recs[:, REC_RECV], recs[:, REC_SRC])
        if dsound is not None:
            pk = np.concatenate([pk, self._pair_key(table[has_dis], recs[has_dis, REC_RECV], recs[has_dis, REC_DIS_SRC])])
        slot = self._lookup(self._pair_keys, self._pair_slots, pk)
        need = slot < 0
        if store.truncate_to is None:
            need |= (slot >= 0) & store._clipped[np.maximum(slot, 0)]
        if need.any():
            self._load_pairs(pk, np.flatnonzero(need))
            changed = True
        return changed

    @staticmethod
    def _records(requests: Sequence[AudioRequest]) -> Optional[bytes]:
        """the packed records of a step, concatenated (None: some request has none - live RIRs - the step takes the walk)"""
        try:
            return b"".join([q.rec for q in requests])
        except TypeError:
            return None

    def _request_tables(self):
        if self._tables is None:
            store = self.engine.store
            ctx = self.engine.context()
            pk, ps = self._pair_keys, self._pair_slots
            if self._use_loader():
                # the library extends the resident-pair arrays in place when it serves a step's pose misses itself
                # (ss_ctx_observe_requests_load): they live at the front of buffers with spare capacity
                n = pk.shape[0]
                if self._pair_buf is None or pk.base is not self._pair_buf[0] or self._pair_buf[0].shape[0] < n + 128:
                    cap = max(2 * n, n + 1024)
                    bk, bs = np.zeros((cap,), np.int64), np.zeros((cap,), np.int64)
                    bk[:n], bs[:n] = pk, ps
                    self._pair_buf = (bk, bs)
                    self._pk, self._ps = bk[:n], bs[:n]
                pk, ps = self._pair_buf
            t = ctx.request_tables(self._sound_keys, self._sound_ids, self._table_keys, self._table_ids, pk, ps,
                                   stale=store._clipped if store.truncate_to is None else None,
                                   last_used=getattr(store, "_batch_of", None))
            t["t"].n_pairs = int(self._pk.shape[0])         # (the buffers are longer than what is resident)
            self._tables = t
        return self._tables

    def _use_loader(self) -> bool:
        """May the library serve pose misses itself?  (a GPU ``RirStore`` with one row per key behind the stock wav reader)"""
        if self._loader_ok is None:
            from .renderer import RirStore, _native_wav
            import functools
            store = self.engine.store
            strict = not (isinstance(self.rir_reader, functools.partial) and self.rir_reader.keywords.get("lenient"))
            self._loader_ok = bool(self.native_miss_path and type(store) is RirStore and store.group == 1 and
                                   store.device.type == "cuda" and hasattr(self.engine, "_sync_context_bank") and
                                   _native_wav(self.rir_reader) and strict)
        return self._loader_ok

    def _miss_loader(self):
        """the store's ss_miss_loader over this resolver's directories and pair buffers, refreshed for the coming call"""
        store = self.engine.store
        bk, bs = self._pair_buf
        if self._loader is None:
            self._loader = store.miss_loader(self._table_dirs, bk, bs, int(self._pk.shape[0]))
        else:
            store.refresh_loader(self._loader, self._table_dirs, bk, bs)
        return self._loader

    def _adopt_loaded(self) -> None:
        """after a call in which the library loaded poses itself: the store's dictionaries and this resolver's views follow"""
        tables = self._tables
        k = self.engine.store.adopt_loaded(self._loader, lambda key: ("ix", key))
        if k:
            # entries the library evicted went through the store's hooks (_evicted: queued + tables dropped) like any other
            # eviction - but their pairs have ALREADY left the sorted arrays inside the call: nothing to drop, the tables stand
            self._evq = []
            self._tables = tables
            n = int(tables["t"].n_pairs)
            self._pk, self._ps = self._pair_buf[0][:n], self._pair_buf[1][:n]
            self.library_loaded += k

    def _columns(self, requests: Sequence[AudioRequest], buf: Optional[bytes] = None):
        """The N requests of a vector step -> unit columns for ``engine.observe_columns`` in numpy, registering what is new:
        sounds and RIR tables never seen before, poses whose RIR file is not resident yet (the only per-request Python).
        Engines with ``observe_requests`` run the same lookups in C++ (``ss_ctx_observe_requests``) and come here only for
        steps that have something to register or load."""
        if buf is None:
            buf = self._records(requests)
            if buf is None:
                return None
        recs = np.frombuffer(buf, np.int64).reshape(-1, REC_N)
        n = recs.shape[0]
        live = recs[:, REC_SILENT] == 0
        has_dis = live & (recs[:, REC_DIS_SOUND] >= 0)
        any_dis = bool(has_dis.any())
        for _ in range(2):
            sound = self._lookup(self._sound_keys, self._sound_ids, recs[:, REC_SOUND])
            table = self._lookup(self._table_keys, self._table_ids, recs[:, REC_TABLE])
            dsound = self._lookup(self._sound_keys, self._sound_ids, recs[:, REC_DIS_SOUND]) if any_dis else None
            new = live & ((sound < 0) | (table < 0))
            if any_dis:
                new |= has_dis & (dsound < 0)
            if not new.any():
                break
            if requests is None:                            # (callers without request objects register names themselves)
                raise KeyError("deferred audio: a record names a sound or RIR directory that was never registered")
            for i in np.flatnonzero(new):                   # first use of a sound / a RIR directory
                q = requests[i]
                if sound[i] < 0 and name_key(q.sound) not in self._key_names:
                    self._learn_sound(q.sound, q.clip)
                if table[i] < 0 and name_key(os.path.dirname(q.rir_key)) not in self._key_names:
                    self._learn_table(q.rir_key)
                if any_dis and has_dis[i] and dsound[i] < 0 and name_key(q.dis_sound) not in self._key_names:
                    self._learn_sound(q.dis_sound, q.dis_clip)
        else:
            raise KeyError("deferred audio: a request names a sound or RIR directory that could not be registered")
        store = self.engine.store
        store.begin_batch()
        tbl = np.where(live, table, 0)
        pk = self._pair_key(tbl, recs[:, REC_RECV], recs[:, REC_SRC])
        dpk = self._pair_key(tbl, recs[:, REC_RECV], recs[:, REC_DIS_SRC]) if any_dis else None
        for attempt in range(2):
            rir = self._lookup(self._pair_keys, self._pair_slots, pk)
            drir = self._lookup(self._pair_keys, self._pair_slots, dpk) if any_dis else None
            used = rir[live & (rir >= 0)]
            if any_dis:
                used = np.concatenate([used, drir[has_dis & (drir >= 0)]])
            store.touch_slots(used)
            miss = live & (rir < 0)
            dmiss = has_dis & (drir < 0) if any_dis else None
            # rows clipped while only 1-s clips existed are reloaded once whole RIRs are needed (RirStore.slot does it)
            if store.truncate_to is None and used.shape[0] and store._clipped[used].any():
                miss = miss | (live & (rir >= 0) & store._clipped[np.maximum(rir, 0)])
                if any_dis:
                    dmiss = dmiss | (has_dis & (drir >= 0) & store._clipped[np.maximum(drir, 0)])
            if not (miss.any() or (any_dis and dmiss.any())):
                break
            if attempt:
                raise KeyError("deferred audio: RIR pairs still missing after loading them")
            self._load_pairs(pk, np.flatnonzero(miss))
            if any_dis:
                self._load_pairs(dpk, np.flatnonzero(dmiss))
        cols = dict(sound=np.where(live, sound, 0), t0=np.where(live, recs[:, REC_T0], 0), rir=np.where(live, rir, -1))
        if any_dis:
            cols.update(dis_sound=np.where(has_dis, dsound, 0), dis_rir=np.where(has_dis, drir, -1))
        return cols

    def resolve(self, requests: Sequence[AudioRequest], want_audiogoal: bool = False, want_spectrogram: bool = True,
                spectrogram_out=None, audiogoal_out=None):
        """-> {"spectrogram": [N,65,T4,2], ("audiogoal": [N,2,sr])} device tensors, one launch for all envs."""
        if any(q.cache_key is not None for q in requests):
            return self._resolve_with_pose_cache(requests, want_audiogoal, want_spectrogram, spectrogram_out, audiogoal_out)
        return self._resolve(requests, want_audiogoal, want_spectrogram, spectrogram_out, audiogoal_out)

    def pose_pool_bytes(self) -> int:
        """HBM held by the pose pool (pose_cache workers): 13.5 KB per cached pose at 16 kHz, 141.5 KB when the audiogoal is
        cached too; rows are freed when an env's simulator drops its caches (scene / sound change, simulator.py:395-397) and
        the pool itself only grows - size it by (poses visited per episode) x (envs)."""
        return int(sum(t.numel() * t.element_size() for t in self._pose_pool.values()))

    def _resolve_with_pose_cache(self, requests, want_audiogoal, want_spectrogram, spectrogram_out, audiogoal_out):
        """pose_cache workers: misses are rendered and their rows kept in a device-side pool under (env, pose); hits are
        row copies out of it (the reference returns the array it cached at that pose, simulator.py:683-686)."""
        import torch
        hits = [i for i, q in enumerate(requests) if q.cache_hit]
        # a hit renders nothing: it rides through the launch as a silent unit (exact zeros, then overwritten)
        reqs = [AudioRequest(env=q.env, kind=q.kind, silent=True, rec=_silent_rec(q.env)) if q.cache_hit else q for q in requests]
        out = self._resolve(reqs, want_audiogoal, want_spectrogram, spectrogram_out, audiogoal_out)
        ctx = getattr(self.engine, "_ctx", None)
        if ctx is not None:
            ctx.join()                                        # (overlap lanes: the pool copies below READ and WRITE the step's rows)
        dev = next(iter(out.values())).device
        maps = self._pose_maps
        miss_rows, miss_slots = [], []
        for i, q in enumerate(requests):
            if q.cache_key is None:
                continue
            m = maps.setdefault(q.env, [q.cache_epoch, {}])
            if m[0] != q.cache_epoch:                           # the worker's simulator dropped its caches
                self._pose_free += list(m[1].values())
                m[0], m[1] = q.cache_epoch, {}
            if not q.cache_hit:
                miss_rows.append(i)
        if self._pose_pool and set(out) != set(self._pose_pool):
            # the pool holds one row per cached pose and OUTPUT; a step that asks for another set (an AudioGoalSensor that
            # appears after the first step) can neither be stored nor served from it
            raise KeyError(f"pose_cache: this step's outputs {sorted(out)} differ from those the pose pool was created with "
                           f"{sorted(self._pose_pool)}: configure the same audio sensors for every step")
        if miss_rows:
            while len(self._pose_free) < len(miss_rows):
                new_cap = max(256, 2 * self._pose_cap)
                for name, t in out.items():
                    grown = torch.empty((new_cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                    if name in self._pose_pool:
                        grown[:self._pose_cap] = self._pose_pool[name]
                    self._pose_pool[name] = grown
                self._pose_free += list(range(new_cap - 1, self._pose_cap - 1, -1))
                self._pose_cap = new_cap
            for i in miss_rows:
                j = self._pose_free.pop()
                maps[requests[i].env][1][requests[i].cache_key] = j
                miss_slots.append(j)
            mi, sj = (torch.as_tensor(v, dtype=torch.long, device=dev) for v in (miss_rows, miss_slots))
            for name, t in out.items():
                self._pose_pool[name].index_copy_(0, sj, t.index_select(0, mi))
        if hits:
            slots = [maps[requests[i].env][1][requests[i].cache_key] for i in hits]
            hi, hj = (torch.as_tensor(v, dtype=torch.long, device=dev) for v in (hits, slots))
            for name, t in out.items():
                if name not in self._pose_pool:
                    raise KeyError(f"pose_cache: {name} was not among the outputs when this pose was first rendered")
                t.index_copy_(0, hi, self._pose_pool[name].index_select(0, hj))
        return out

    def _resolve(self, requests: Sequence[AudioRequest], want_audiogoal: bool = False, want_spectrogram: bool = True,
                 spectrogram_out=None, audiogoal_out=None):
        buf = self._records(requests) if self.columns_ok else None
        if buf is not None:
            return self.resolve_records(buf, len(requests), requests, want_audiogoal, want_spectrogram, spectrogram_out,
                                        audiogoal_out)
        if self.columns_ok:                                  # numbered live RIRs (SoundSpaces 2.0 workers): columns, no walk
            cols = self._live_columns(requests)
            if cols is not None:
                return self._observe_columns(cols, len(requests), want_audiogoal, want_spectrogram, spectrogram_out, audiogoal_out)
        self.walk_steps += 1
        return self.engine.observe(self.units(requests), want_audiogoal=want_audiogoal or audiogoal_out is not None,
                                   want_spectrogram=want_spectrogram, spectrogram_out=spectrogram_out,
                                   audiogoal_out=audiogoal_out)

    def _observe_columns(self, cols, n, want_audiogoal, want_spectrogram, spectrogram_out, audiogoal_out):
        import torch
        r = self.engine.renderer
        want_audiogoal = want_audiogoal or audiogoal_out is not None
        if want_spectrogram and spectrogram_out is None:
            spectrogram_out = torch.empty((n,) + tuple(r.spectrogram_shape), dtype=torch.float32, device=r.device)
        if want_audiogoal and audiogoal_out is None:
            audiogoal_out = torch.empty((n, 2, r.out_len), dtype=torch.float32, device=r.device)
        self.engine.observe_columns(cols, spectrogram_out=spectrogram_out if want_spectrogram else None,
                                    audiogoal_out=audiogoal_out if want_audiogoal else None)
        self.live_steps += 1
        out = {}
        if want_spectrogram:
            out["spectrogram"] = spectrogram_out
        if want_audiogoal:
            out["audiogoal"] = audiogoal_out
        return out

    def resolve_records(self, buf: bytes, n: int, requests: Optional[Sequence[AudioRequest]] = None,
                        want_audiogoal: bool = False, want_spectrogram: bool = True, spectrogram_out=None, audiogoal_out=None):
        """One step from its packed records (n x REC_N int64 words as bytes; `requests` only serves first-use registration
        of sounds / RIR directories and may be None when the caller registers them itself - the in-process
        ``VectorAudioObserver`` does)."""
        import torch
        r = self.engine.renderer
        want_audiogoal = want_audiogoal or audiogoal_out is not None
        if want_spectrogram and spectrogram_out is None:
            spectrogram_out = torch.empty((n,) + tuple(r.spectrogram_shape), dtype=torch.float32, device=r.device)
        if want_audiogoal and audiogoal_out is None:
            audiogoal_out = torch.empty((n, 2, r.out_len), dtype=torch.float32, device=r.device)
        sg, ag = (spectrogram_out if want_spectrogram else None), (audiogoal_out if want_audiogoal else None)
        done = False
        if hasattr(self.engine, "observe_requests"):       # lookups + planner + launch in one C call
            store = self.engine.store
            ticking = hasattr(store, "_batch_of")
            if ticking:                                    # the store's LRU clock: the C lookups stamp the rows they use
                store.begin_batch()
            if getattr(self.engine, "spectral_max_units", 0):
                # (engines that pick the bank form per step: does this step carry distractor terms?  renderer.AudioEngine)
                self.engine._req_has_distractor = bool((np.frombuffer(buf, np.int64).reshape(n, REC_N)[:, REC_DIS_SOUND] >= 0).any())
            for attempt in range(3):
                tables = self._request_tables()
                if ticking:
                    tables["t"].tick = store._batch
                if self._use_loader() and not self._evq:
                    n_miss = self.engine.observe_requests(buf, n, tables, spectrogram_out=sg, audiogoal_out=ag,
                                                          loader=self._miss_loader())
                    if self._loader["s"].n_loaded:
                        self._adopt_loaded()
                else:
                    n_miss = self.engine.observe_requests(buf, n, tables, spectrogram_out=sg, audiogoal_out=ag)
                if n_miss == 0:
                    done = True
                    break
                # a step with something new (a pose whose RIR file is not resident, a first sound / RIR directory): only the
                # reported requests are looked at - files read by the library's reader in one call - and the C call runs again
                if attempt == 2 or not self._serve_misses(buf, n, requests, self.engine._req_miss["buf"][:n_miss]):
                    break
            self.native_steps += done
            self.miss_steps += bool(done and attempt)
        if not done:                                        # something to register / load (or an engine without the C path)
            self.engine.observe_columns(self._columns(requests, buf), spectrogram_out=sg, audiogoal_out=ag)
        self.column_steps += 1
        out = {}
        if want_spectrogram:
            out["spectrogram"] = spectrogram_out
        if want_audiogoal:
            out["audiogoal"] = audiogoal_out
        return out

    def resolve_observations(self, observations: Sequence[dict], rollouts=None, replace: bool = True):
        """Replacement for the audio half of ``batch_obs`` (ss_baselines/common/utils.py:126-153): `observations` is
        the list of per-env dicts the vector env returned; entries under 'spectrogram' / 'audiogoal' that are
        ``AudioRequest`` s are rendered in one launch (into the rollout rows of the next ``insert()`` when `rollouts` is
        given) and REPLACED, per env, by their device tensors (views of the batch), so the dicts can go on to
        ``batch_obs`` unchanged (``replace=False``: trainers that take the audio from the returned batch / the rollout
        rows leave the requests in the dicts and save the N views).  Returns the batched tensors {uuid: [N, ...]}."""
        keys = [k for k in ("spectrogram", "audiogoal") if observations and isinstance(observations[0].get(k), AudioRequest)]
        if not keys:
            return {}
        reqs = [obs[keys[0]] for obs in observations]
        slots = rollouts.next_observation_slots([k for k in keys if k in rollouts.observations]) if rollouts is not None else {}
        if len(keys) == 2 and any(obs["audiogoal"].t0 != obs["spectrogram"].t0 or obs["audiogoal"].silent != obs["spectrogram"].silent
                                  for obs in observations):
            # HAS_DISTRACTOR_SOUND: the reference computes per sensor read (simulator.py:679-681) - the two sensors of a step hear
            # consecutive seconds of a multi-second sound.  One launch per sensor, each from its own requests.
            # (the sensor that was read FIRST carries the clips a worker sends once: its requests are resolved first)
            ag_reqs = [obs["audiogoal"] for obs in observations]
            ag_first = any(q.clip is not None or q.dis_clip is not None for q in ag_reqs)
            out = {}
            for which in (("audiogoal", "spectrogram") if ag_first else ("spectrogram", "audiogoal")):
                if which == "spectrogram":
                    out.update(self.resolve(reqs, want_audiogoal=False, want_spectrogram=True, spectrogram_out=slots.get("spectrogram")))
                else:
                    out.update(self.resolve(ag_reqs, want_audiogoal=True, want_spectrogram=False, audiogoal_out=slots.get("audiogoal")))
        else:
            out = self.resolve(reqs, want_audiogoal="audiogoal" in keys, want_spectrogram="spectrogram" in keys,
                               spectrogram_out=slots.get("spectrogram"), audiogoal_out=slots.get("audiogoal"))
        if replace:
            for k in keys:
                for obs, row in zip(observations, out[k].unbind(0)):      # ONE call makes the N row views
                    obs[k] = row
        return {k: out[k] for k in keys}

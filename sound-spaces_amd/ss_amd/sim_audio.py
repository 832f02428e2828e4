"""Simulator-side audio of SoundSpaces on the HIP path.

``HipSimAudio`` re-implements, for one simulator instance, the three audio methods the task sensors call

    sim.get_current_audiogoal_observation()                    soundspaces/simulator.py:678-688
    sim.get_current_spectrogram_observation(audiogoal2spec)    soundspaces/simulator.py:690-701
    sim._compute_audiogoal()                                   soundspaces/simulator.py:608-666

with the arithmetic done by the batched renderer (one unit per call in this *eager* mode).  ``attach()``
installs them on a live ``SoundSpacesSim`` so the reference's own sensors keep working unchanged.  The two memo
caches live on the simulator object under the reference's attribute names, so the simulator's own
``reconfigure`` (simulator.py:395-397) keeps clearing them.

``HipContinuousSimAudio`` / ``attach_continuous()`` do the same for ``ContinuousSoundSpacesSim``
(soundspaces/continuous_simulator.py:413-462: live RIR, 0.25-s steps, cross-fade, no caches).

``VectorAudioObserver`` is the *batched* mode: it collects one request per env of an in-process vector env
(ss_baselines/common/sync_vector_env.py) and renders all of them in one launch straight into a device tensor —
what replaces the per-env sensor call + ``batch_obs`` (ss_baselines/common/utils.py:126-153) for the audio keys.
"""
from __future__ import annotations

import logging
import os
from typing import Callable, List, Optional

import numpy as np

from .renderer import UnitRequest


def wav_rir_reader(path: str, lenient: bool = False) -> Optional[np.ndarray]:
    """simulator.py:615-624: float32 [L,2] from a wav, or None (-> zero RIR) when the file is not a readable wav
    (``ValueError``, the only exception the reference catches, :617-620) or empty.  A MISSING file propagates, as it
    does in the reference: a wrong ``binaural_rir_dir`` must not turn into a run of all-zero observations.
    ``lenient=True`` also maps FileNotFoundError / OSError to the zero RIR (partial RIR sets)."""
    from scipy.io import wavfile
    try:
        _, rir = wavfile.read(path)
    except ValueError:
        logging.warning("%s file is not readable", path)
        return None
    except OSError:
        if not lenient:
            raise
        logging.warning("%s file is missing", path)
        return None
    if len(rir) == 0:
        logging.debug("Empty RIR file at %s", path)
        return None
    return np.asarray(rir, dtype=np.float32)


# eager observations: the kernels write their outputs straight into the pinned host buffer (device-mapped under ROCm) instead
# of a device buffer that a copy-engine job then moves (scripts/prof_eager.py A/B: see profiles/r4/NOTES.md)
EAGER_DIRECT_HOST = True


def same_rir(a: np.ndarray, b: np.ndarray) -> bool:
    """a and b hold the same RIR.  A thin strided sample rejects different arrays first (live RIRs differ everywhere; comparing
    72 KB in full against each held row was ~30 us per env and step), the full comparison only confirms a match."""
    if a is b:
        return True
    if a.shape != b.shape:
        return False
    n = a.shape[0]
    if n > 64:
        step = n // 32
        if not np.array_equal(a[::step], b[::step]):
            return False
    return bool(np.array_equal(a, b))


def _render_to_host(backend, req, want_spectrogram: bool, want_audiogoal: bool = True):
    """One unit through the engine and back to the host as (audiogoal [2, sr] or None, spectrogram or None), numpy arrays owned
    by the caller (the reference's are too: they end up in the simulator's caches).  Both outputs land in ONE pinned buffer
    (written by the kernels themselves, or as one async copy) - two ``.cpu()`` calls, i.e. two synchronising copies through
    pageable memory, were 56 of the 92 us of an eager observation.  want_audiogoal False: the waveform never leaves the CU."""
    eng, sr = backend.engine, backend.sr
    if not hasattr(eng, "renderer"):                             # an engine without device buffers of its own (test doubles)
        out = eng.observe([req], want_audiogoal=True, want_spectrogram=want_spectrogram)
        return (out["audiogoal"][0].cpu().numpy() if want_audiogoal else None,
                out["spectrogram"][0].cpu().numpy() if want_spectrogram else None)
    st = getattr(backend, "_stage", None)
    if st is None:
        import torch
        from . import ops
        from .planning import spectrogram_shape
        shp = spectrogram_shape(sr)
        n_ag, n_sg = 2 * sr, shp[0] * shp[1] * shp[2]
        dbuf = torch.empty(n_ag + n_sg, dtype=torch.float32, device=eng.renderer.device)
        hbuf = torch.empty(n_ag + n_sg, dtype=torch.float32).pin_memory()
        st = backend._stage = (dbuf, hbuf, hbuf.numpy(), shp, n_ag, n_sg, torch, ops, dbuf[:0],
                               torch.ops.ss_hip.eager_obs if ops.NATIVE_OPS else None)
    dbuf, hbuf, h, shp, n_ag, n_sg, torch, ops, dbuf0, eager_op = st
    n = n_ag + n_sg if want_spectrogram else n_ag
    if ops.NATIVE_OPS and hasattr(eng, "context"):
        # ONE C++ dispatch (csrc/ss_torch_ops.cpp::eager_obs): the library's planner + window cache + launch
        # (ss_ctx_observe on a one-unit step), one async copy of both outputs into pinned memory, stream synchronise
        ctx = eng._sync_context_bank()
        wrap = 1 if req.wrap is None else int(bool(req.wrap))
        eager_op(ctx.handle, int(req.sound), int(req.t0), int(req.rir), int(req.dis_sound), int(req.dis_rir),
                 int(req.last_rir), wrap, wrap if req.last_wrap is None else int(bool(req.last_wrap)),
                 dbuf0 if EAGER_DIRECT_HOST else dbuf, hbuf, sr, want_spectrogram, want_audiogoal)
    else:
        eng.observe([req], want_audiogoal=True, want_spectrogram=want_spectrogram, audiogoal_out=dbuf[:n_ag].view(1, 2, sr),
                    spectrogram_out=dbuf[n_ag:].view((1,) + shp) if want_spectrogram else None)
        hbuf[:n].copy_(dbuf[:n], non_blocking=True)
        torch.cuda.current_stream(dbuf.device).synchronize()     # (an event record + synchronize measured 10 us slower)
    return (h[:n_ag].reshape(2, sr).copy() if want_audiogoal else None,
            h[n_ag:n].reshape(shp).copy() if want_spectrogram else None)


class HipSimAudio:
    _ids = __import__("itertools").count()

    def __init__(self, sim, engine, rir_reader: Callable[[str], Optional[np.ndarray]] = wav_rir_reader,
                 lazy_audiogoal: bool = True):
        """engine: ss_amd.renderer.AudioEngine (or any object with source_id / rir_slot / observe).

        lazy_audiogoal (default since round 6; False = fetch both outputs with every launch, as the reference computes them;
        it pays for tasks whose only audio sensor is the SpectrogramSensor - av_nav's default): a spectrogram
        request computes and fetches the spectrogram ONLY (the waveform stays on the CU: no 128 KB over PCIe, no host copy);
        the simulator's ``_audiogoal_cache`` is then filled on demand - an AudioGoalSensor read of a pose whose spectrogram
        was rendered re-renders the waveform from the SAME request (same clip window, ``_audio_index`` not advanced again:
        the reference computes both at once, simulator.py:690-701) - and from the first such read on both are fetched
        together again, as without the option."""
        self.sim = sim
        self.engine = engine
        self.rir_reader = rir_reader
        self.lazy_audiogoal = bool(lazy_audiogoal)
        self._ag_wanted = False                      # an audiogoal read has been seen: fetch both outputs per launch
        self._pending = {}                           # pose -> (spectrogram it belongs to, request) awaiting an audiogoal read
        self._refs = []
        self._paths = {}                             # (rir dir, azimuth, receiver, source) -> the RIR file path
        self._env_id = next(HipSimAudio._ids)        # stable key of this env's live RIR row (USE_RENDERED_OBSERVATIONS False)
        for name in ("_audiogoal_cache", "_spectrogram_cache"):
            if not isinstance(getattr(sim, name, None), dict):
                setattr(sim, name, {})

    # ---- state -> request ------------------------------------------------------------------------------
    @property
    def sr(self) -> int:
        return int(self.sim.config.AUDIO.RIR_SAMPLING_RATE)

    def _rir_slot(self, source_index, from_file: bool = False) -> int:
        sim = self.sim
        if from_file or sim.config.USE_RENDERED_OBSERVATIONS:
            key = (sim.binaural_rir_dir, sim.azimuth_angle, sim._receiver_position_index, source_index)
            path = self._paths.get(key)                  # (the string is built once per pose: 2 us of a 45-us eager call)
            if path is None:
                if len(self._paths) >= 1 << 16:
                    self._paths.clear()
                path = self._paths[key] = os.path.join(key[0], str(key[1]), "{}_{}.wav".format(key[2], source_index))  # :615-616, :650-651
            self._refs.append(path)
            file_slot = getattr(self.engine, "rir_file_slot", None)
            if file_slot is not None:                        # (misses of the stock reader: the library's own wav reader)
                return file_slot(path, self.rir_reader)
            return self.engine.rir_slot(path, lambda: self.rir_reader(path))
        # habitat_sim audio sensor: a fresh RIR every step (:626) -> this env's live slot, re-uploaded.  The key is a
        # counter drawn at attach time: id(sim) can be handed to another simulator once this one is collected
        rir = np.transpose(np.array(sim._sim.get_sensor_observations()["audio_sensor"]))
        self._refs.append(rir)
        return self.engine.rir_slot(("live", self._env_id), lambda: rir, refresh=True)

    def _reslot(self, req: UnitRequest, refs) -> None:
        """A request kept for later (lazy_audiogoal) names RAW store slots: by the time it is rendered again the store may have
        given them to other keys (LRU), or - live RIRs - this env's one row holds a later step's RIR.  Resolve its
        RIRs again from what they WERE: the file paths / the live RIR array of the step the request was made in."""
        def slot(ref):
            if isinstance(ref, str):
                file_slot = getattr(self.engine, "rir_file_slot", None)
                return file_slot(ref, self.rir_reader) if file_slot is not None else self.engine.rir_slot(ref, lambda: self.rir_reader(ref))
            return self.engine.rir_slot(("live", self._env_id), lambda: ref, refresh=True)
        if refs:
            req.rir = slot(refs[0])
        if len(refs) > 1 and req.dis_rir is not None and req.dis_rir >= 0:
            req.dis_rir = slot(refs[1])

    def unit_request(self) -> UnitRequest:
        """The request _compute_audiogoal would serve right now.  Advances ``_audio_index`` exactly where the
        reference does (:634-635: multi-second sounds only, not when silent)."""
        sim, sr = self.sim, self.sr
        self._refs = []                                  # what the request's RIR slots were resolved from (see _reslot)
        if sim._episode_step_count > sim._duration:                                                  # :610
            return UnitRequest(silent=True)
        clip = sim.current_source_sound
        sound = self.engine.source_id(sim._current_sound, clip)
        rir = self._rir_slot(sim._source_position_index)
        if clip.shape[0] == sr:                                                                      # :629
            t0 = 0
        else:
            index = sim._audio_index                                                                 # :634
            sim._audio_index = (sim._audio_index + 1) % sim._audio_length                            # :635
            t0 = index * sr
        req = UnitRequest(sound=sound, t0=t0, rir=rir)
        if sim.config.AUDIO.HAS_DISTRACTOR_SOUND:                                                    # :649
            dclip = sim._source_sound_dict[sim._current_distractor_sound]
            req.dis_sound = self.engine.source_id(sim._current_distractor_sound, dclip)
            req.dis_rir = self._rir_slot(sim._distractor_position_index, from_file=True)   # always the wav file (:650-658)
        return req

    def _joint_index(self):
        sim = self.sim
        return (sim._source_position_index, sim._receiver_position_index, sim.azimuth_angle)       # :683

    # ---- the reference API ----------------------------------------------------------------------------------
    def _compute(self, want_spectrogram: bool, want_audiogoal: bool = True, keep=None):
        if hasattr(self.engine, "begin_batch"):
            self.engine.begin_batch()
        req = self.unit_request()
        if keep is not None:
            keep.append(req)
        if req.silent:
            # simulator.py:610-612: np.zeros((2, sr)) - FLOAT64, and so is the spectrogram nav.py:86-100 makes of it; no launch
            from .planning import spectrogram_shape
            return np.zeros((2, self.sr)), (np.zeros(spectrogram_shape(self.sr)) if want_spectrogram else None)
        return _render_to_host(self, req, want_spectrogram, want_audiogoal)

    def get_current_audiogoal_observation(self):
        sim = self.sim
        self._ag_wanted = True
        if sim.config.AUDIO.HAS_DISTRACTOR_SOUND:                                                    # :679-681
            return self._compute(False)[0]
        key = self._joint_index()
        if key not in sim._audiogoal_cache:
            pend = self._pending.pop(key, None) if self._pending else None
            if pend is not None and sim._spectrogram_cache.get(key) is pend[0]:
                # lazy_audiogoal: this pose's spectrogram was rendered without its waveform; the waveform of THAT request
                # (the simulator has not dropped its caches since: the spectrogram is still the cached one)
                if hasattr(self.engine, "begin_batch"):
                    self.engine.begin_batch()
                self._reslot(pend[1], pend[2])
                sim._audiogoal_cache[key] = _render_to_host(self, pend[1], False)[0]
            else:
                sim._audiogoal_cache[key] = self._compute(False)[0]
        return sim._audiogoal_cache[key]

    def get_current_spectrogram_observation(self, audiogoal2spectrogram=None):
        """``audiogoal2spectrogram`` is accepted for signature compatibility; when it is (or wraps) this package's
        ``SpectrogramSensor.compute_spectrogram`` (or is None) the fused kernel result is used, any other callable is
        applied to the audiogoal on the host, exactly like the reference would."""
        sim = self.sim
        foreign = audiogoal2spectrogram is not None and not getattr(audiogoal2spectrogram, "_ss_hip_fused", False)
        if sim.config.AUDIO.HAS_DISTRACTOR_SOUND:                                                    # :691-693
            if foreign:
                return audiogoal2spectrogram(self.get_current_audiogoal_observation())
            return self._compute(True, want_audiogoal=not (self.lazy_audiogoal and not self._ag_wanted))[1]
        key = self._joint_index()
        if key not in sim._spectrogram_cache:
            if foreign:
                sim._spectrogram_cache[key] = audiogoal2spectrogram(self.get_current_audiogoal_observation())
            elif key in sim._audiogoal_cache:        # waveform already cached (AudioGoalSensor ran first)
                from .sensors import SpectrogramSensor
                sim._spectrogram_cache[key] = SpectrogramSensor.compute_spectrogram(sim._audiogoal_cache[key])
            elif self.lazy_audiogoal and not self._ag_wanted:
                if self._pending and not sim._spectrogram_cache:      # the simulator dropped its caches (:395-397): so do we
                    self._pending.clear()
                keep = []
                ag, sg = self._compute(True, want_audiogoal=False, keep=keep)
                sim._spectrogram_cache[key] = sg
                if ag is not None:                   # (silent step: the zeros cost nothing)
                    sim._audiogoal_cache[key] = ag
                else:
                    if len(self._pending) > 4096:    # (the simulator clears its caches per episode; this map follows lazily)
                        self._pending.clear()
                    self._pending[key] = (sg, keep[0], list(self._refs))
            else:                                    # one fused launch fills both caches
                ag, sg = self._compute(True)
                sim._audiogoal_cache[key] = ag
                sim._spectrogram_cache[key] = sg
        return sim._spectrogram_cache[key]


def attach(sim, engine, rir_reader=wav_rir_reader, lazy_audiogoal: bool = True) -> HipSimAudio:
    """Install the HIP audio path on a live SoundSpacesSim: the task sensors (the reference's or ss_amd's) keep
    calling ``sim.get_current_*_observation`` and now reach the GPU renderer.  lazy_audiogoal: see ``HipSimAudio``."""
    backend = HipSimAudio(sim, engine, rir_reader, lazy_audiogoal)
    sim.get_current_audiogoal_observation = backend.get_current_audiogoal_observation
    sim.get_current_spectrogram_observation = backend.get_current_spectrogram_observation
    sim._compute_audiogoal = lambda: backend._compute(False)[0]
    sim._ss_hip_audio = backend
    return backend


class HipContinuousSimAudio:
    """Audio of ``ContinuousSoundSpacesSim`` (SoundSpaces 2.0; the reference's default DD-PPO mode,
    ss_baselines/av_nav/single_node.sh:14) on the HIP path:

        sim._compute_audiogoal()                               soundspaces/continuous_simulator.py:413-426
        sim._convolve_with_rir(rir)                            soundspaces/continuous_simulator.py:428-456
        sim.get_current_audiogoal_observation()                :458-459   (uncached)
        sim.get_current_spectrogram_observation(fn)            :461-462   (uncached)

    State read, exactly what the reference reads: ``_episode_step_count`` / ``_duration`` (silence), the live RIR
    ``_prev_sim_obs["audio_sensor"]`` ([2][L] from the ray tracer, transposed at :419), ``_last_rir`` and
    ``config.AUDIO.CROSSFADE`` (:422-424), ``_current_sample_index`` (advanced by the simulator's own ``step``, :389-390),
    ``config.STEP_TIME``, ``current_source_sound`` (1-s clips already tiled x3 by ``_load_single_source_sound``,
    :408-410).  The engine must be an SS2.0 one: ``AudioEngine(sr, step_time=STEP_TIME, wrap=True)``.

    Live RIRs change every step, so each env owns two bank slots used alternately: the RIR uploaded as "current" at
    step k is the "previous" RIR of step k+1 and is recognised by content (``_last_rir`` is a fresh transposed copy in
    the simulator), so a cross-faded step uploads ONE new RIR, not two, and the two sensors of a step (audiogoal and
    spectrogram both call ``_compute_audiogoal`` in the reference) upload it once."""

    _ids = __import__("itertools").count()

    def __init__(self, sim, engine):
        self.sim, self.engine = sim, engine
        self._env_id = next(HipContinuousSimAudio._ids)   # stable key of this env's live rows (id(sim) can be reused after GC)
        self._held = [None, None]             # the arrays living in this env's two live slots
        self._slots = [-1, -1]
        self._turn = 0
        r = getattr(engine, "renderer", None)
        if r is not None:
            want = int(self.sr * float(sim.config.STEP_TIME))
            if r.n_valid != want or not r.wrap:
                raise ValueError(f"continuous simulator needs AudioEngine(sr, step_time={sim.config.STEP_TIME}, wrap=True); "
                                 f"got n_valid={r.n_valid}, wrap={r.wrap}")

    @property
    def sr(self) -> int:
        return int(self.sim.config.AUDIO.RIR_SAMPLING_RATE)

    def _live_slot(self, rir: np.ndarray, avoid: int = -1) -> int:
        """Bank slot holding ``rir`` ([L, 2] wav layout); uploads into this env's other live slot unless one of the
        two already holds exactly this array."""
        for k in (0, 1):
            h = self._held[k]
            if h is not None and same_rir(h, rir):
                # through the store even on a content match: LRU touch, batch guard, re-upload if the slot was evicted
                self._slots[k] = self.engine.rir_slot(("live", self._env_id, k), lambda: rir, refresh=False)
                return self._slots[k]
        k = self._turn
        if self._slots[k] == avoid and avoid >= 0:
            k ^= 1
        self._turn = k ^ 1
        self._held[k] = rir
        self._slots[k] = self.engine.rir_slot(("live", self._env_id, k), lambda: rir, refresh=True)
        return self._slots[k]

    def unit_request(self) -> UnitRequest:
        sim = self.sim
        if sim._episode_step_count > sim._duration:                                                  # :415
            return UnitRequest(silent=True)
        clip = sim.current_source_sound
        sound = self.engine.source_id(sim._current_sound, clip)
        rir = np.transpose(np.asarray(sim._prev_sim_obs["audio_sensor"], dtype=np.float32))         # :419  [L, 2]
        index = int(sim._current_sample_index)                                                       # :432
        cur = self._live_slot(rir)
        # :433 early branch (index - L < 0): source[:index+num_sample], zeros past the clip end; steady branch wraps
        req = UnitRequest(sound=sound, t0=index, rir=cur, wrap=index - rir.shape[0] >= 0)
        last = getattr(sim, "_last_rir", None)
        if sim.config.AUDIO.CROSSFADE and last is not None:                                          # :422
            last = np.asarray(last, dtype=np.float32)
            req.last_rir = self._live_slot(last, avoid=cur)
            req.last_wrap = index - last.shape[0] >= 0
        return req

    def _compute(self, want_spectrogram: bool):
        if hasattr(self.engine, "begin_batch"):
            self.engine.begin_batch()
        return _render_to_host(self, self.unit_request(), want_spectrogram)

    def get_current_audiogoal_observation(self):                                                     # :458-459
        return self._compute(False)[0]

    def get_current_spectrogram_observation(self, audiogoal2spectrogram=None):                       # :461-462
        if audiogoal2spectrogram is not None and not getattr(audiogoal2spectrogram, "_ss_hip_fused", False):
            return audiogoal2spectrogram(self.get_current_audiogoal_observation())
        return self._compute(True)[1]


def attach_continuous(sim, engine) -> HipContinuousSimAudio:
    """Install the HIP audio path on a live ``ContinuousSoundSpacesSim``; the task sensors keep calling
    ``sim.get_current_*_observation``."""
    backend = HipContinuousSimAudio(sim, engine)
    sim.get_current_audiogoal_observation = backend.get_current_audiogoal_observation
    sim.get_current_spectrogram_observation = backend.get_current_spectrogram_observation
    sim._compute_audiogoal = lambda: backend._compute(False)[0]
    sim._ss_hip_audio = backend
    return backend


class VectorAudioObserver:
    """Batched mode: one launch per vector step for all envs of this process."""

    def __init__(self, engine, backends: List[HipSimAudio], want_audiogoal: bool = False, want_intensity: bool = False):
        """want_intensity: also return av_wan's ``intensity`` [N, 1] (ss_baselines/av_wan/avwan_sensors.py:91-100),
        reduced on the device from the batch's audiogoal."""
        self.engine, self.backends = engine, backends
        self.want_audiogoal, self.want_intensity = want_audiogoal or want_intensity, want_intensity
        self._rec = None                          # record path (see _record_state): None = undecided, False = not applicable
        self.record_steps = self.walk_steps = 0

    # ---- record path: the step's state read with C-level attribute getters, packed into the int64 records of
    # ss_amd.deferred and handed to the C++ context in ONE call (lookups, planner, launch: ss_ctx_observe_requests) ----
    def _record_state(self):
        """SoundSpaces 1.0 simulators on RIR files, one reader, an engine with the column surface (AudioEngine, rir_group=1):
        per step the envs are walked by ``operator.attrgetter`` (one C call per simulator) instead of ``unit_request()``
        (a Python method, a UnitRequest, two store lookups and a Python planner row per env: 3.6 us per env)."""
        import operator
        from .deferred import DeferredResolver
        bs = self.backends
        if not bs or not all(type(b) is HipSimAudio for b in bs):
            return False
        cfgs = [b.sim.config for b in bs]
        if not all(c.USE_RENDERED_OBSERVATIONS for c in cfgs) or any(b.rir_reader != bs[0].rir_reader for b in bs):
            return False
        dis = {bool(c.AUDIO.HAS_DISTRACTOR_SOUND) for c in cfgs}
        if len(dis) != 1 or len({int(c.AUDIO.RIR_SAMPLING_RATE) for c in cfgs}) != 1:
            return False
        try:
            res = DeferredResolver(self.engine, rir_reader=bs[0].rir_reader, fast=True)
        except ValueError:
            return False
        attrs = ["_episode_step_count", "_duration", "_current_sound", "_receiver_position_index", "_source_position_index",
                 "azimuth_angle", "binaural_rir_dir"]
        if True in dis:
            attrs += ["_current_distractor_sound", "_distractor_position_index"]
        return dict(res=res, sims=[b.sim for b in bs], getter=operator.attrgetter(*attrs), dis=True in dis, sr=bs[0].sr,
                    snd={}, tbl={})

    def _learn_sounds(self, st, names) -> None:
        from .deferred import name_key
        res, sr = st["res"], st["sr"]
        for sim, nm in zip(st["sims"], names):
            if nm in st["snd"]:
                continue
            clip = sim._source_sound_dict[nm]
            key = name_key(nm)
            if key not in res._key_names:
                res._learn_sound(nm, np.ascontiguousarray(clip, dtype=np.float32))
            st["snd"][nm] = (key, int(np.shape(clip)[0]) != sr)

    def _observe_records(self, st, spectrogram_out, audiogoal_out):
        from .deferred import (REC_DIS_SOUND, REC_DIS_SRC, REC_ENV, REC_N, REC_RECV, REC_SILENT, REC_SOUND, REC_SRC, REC_T0,
                               REC_TABLE, name_key)
        sims, res, sr = st["sims"], st["res"], st["sr"]
        n = len(sims)
        cols = list(zip(*map(st["getter"], sims)))
        names = cols[2]
        snd = st["snd"]
        try:
            info = [snd[nm] for nm in names]
        except KeyError:
            self._learn_sounds(st, names)
            info = [snd[nm] for nm in names]
        pairs = list(zip(cols[6], cols[5]))
        tbl = st["tbl"]
        try:
            tkeys = [tbl[p] for p in pairs]
        except KeyError:
            for p in pairs:
                if p not in tbl:
                    d = os.path.join(p[0], str(p[1]))                              # simulator.py:615-616
                    if name_key(d) not in res._key_names:
                        res._learn_table(os.path.join(d, "_"))
                    tbl[p] = name_key(d)
            tkeys = [tbl[p] for p in pairs]
        recs = np.zeros((n, REC_N), np.int64)
        silent = np.asarray(cols[0], np.int64) > np.asarray(cols[1], np.int64)      # simulator.py:610
        keys, multi = zip(*info)
        recs[:, REC_SILENT] = silent
        recs[:, REC_SOUND] = keys
        recs[:, REC_TABLE] = tkeys
        recs[:, REC_RECV] = cols[3]
        recs[:, REC_SRC] = cols[4]
        recs[:, REC_DIS_SOUND] = -1
        recs[:, REC_ENV] = np.arange(n)
        if True in multi:                                                           # multi-second clips: simulator.py:634-635
            for i in np.flatnonzero(np.asarray(multi) & ~silent):
                sim = sims[i]
                idx = sim._audio_index
                sim._audio_index = (idx + 1) % sim._audio_length
                recs[i, REC_T0] = idx * sr
        if st["dis"]:                                                               # simulator.py:649-664
            dnames = cols[7]
            try:
                dinfo = [snd[nm] for nm in dnames]
            except KeyError:
                self._learn_sounds(st, dnames)
                dinfo = [snd[nm] for nm in dnames]
            recs[:, REC_DIS_SOUND] = [k for k, _ in dinfo]
            recs[:, REC_DIS_SRC] = cols[8]
        if silent.any():                                                            # (as DeferredSimAudio's silent record)
            recs[silent, REC_SOUND:REC_DIS_SOUND] = 0
            recs[silent, REC_DIS_SOUND] = -1
            recs[silent, REC_DIS_SRC] = 0
        return res.resolve_records(recs.tobytes(), n, None, self.want_audiogoal or audiogoal_out is not None, True,
                                   spectrogram_out, audiogoal_out)

    def observe(self, spectrogram_out=None, audiogoal_out=None):
        """-> {"spectrogram": device tensor [N,65,T4,2], ("audiogoal": [N,2,sr])}; cache-free (every env renders its
        current pose), i.e. the reference's HAS_DISTRACTOR_SOUND / continuous behaviour."""
        if self._rec is None:
            self._rec = self._record_state()
        if self._rec:
            out = self._observe_records(self._rec, spectrogram_out, audiogoal_out)
            self.record_steps += 1
        else:
            if hasattr(self.engine, "begin_batch"):
                self.engine.begin_batch()             # no RIR slot of this step may be evicted by another env of it
            units = [b.unit_request() for b in self.backends]
            out = self.engine.observe(units, want_audiogoal=self.want_audiogoal or audiogoal_out is not None,
                                      want_spectrogram=True, spectrogram_out=spectrogram_out, audiogoal_out=audiogoal_out)
            self.walk_steps += 1
        if self.want_intensity:
            from .sensors import Intensity
            out["intensity"] = Intensity.compute_intensity(out["audiogoal"], 150).reshape(-1, 1)
        return out

    def observe_into(self, rollouts):
        """Render this vector step straight into the rollout rows the next `rollouts.insert()` fills
        (`observations[sensor][step + 1]`, ss_baselines/common/rollout_storage.py:89-92): no per-env arrays, no
        `batch_obs` stack, no H2D copy.  `rollouts` is an `ss_amd.rollout.RolloutStorage`; returns the slots
        (a `DeviceObservations`) to be merged with the other sensors' batch and passed to `insert()`, which
        recognises them and skips the copy."""
        names = [s for s in ("spectrogram", "audiogoal") if s in rollouts.observations]
        slots = rollouts.next_observation_slots(names)
        self.observe(spectrogram_out=slots.get("spectrogram"), audiogoal_out=slots.get("audiogoal"))
        return slots

"""BatchedAudioRenderer — the batched, device-resident replacement for the per-env, per-step audio code of
``SoundSpacesSim`` (soundspaces/simulator.py:608-701) and ``ContinuousSoundSpacesSim``
(soundspaces/continuous_simulator.py:413-462).

One *unit* = one (env, rotation) observation.  The renderer owns

* a source bank: every mono clip once in HBM (the reference keeps ``_source_sound_dict``, simulator.py:595-600),
* an RIR bank: float32 planar ``[R, 2, cap]`` rows, zero padded, resident in HBM (the reference re-reads a wav
  per step, simulator.py:615-618),
* a cache of source-window spectra keyed ``(sound, t0, wrap)`` (the reference recomputes the source FFT inside
  every ``fftconvolve`` call),

and turns N unit requests into one kernel launch that writes ``[N, 65, T4, 2]`` spectrograms (and optionally the
``[N, 2, sr]`` waveforms) on the caller's stream.  No host sync, no CPU arithmetic: if the HIP library is absent the
constructor raises.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from . import planning as P


class SourceBank:
    """Mono source clips, float32, already resampled to the simulator rate (librosa.load(sr=...) upstream)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.names: Dict[str, int] = {}
        self.lengths: List[int] = []
        self.offsets: List[int] = []
        self._host: List[np.ndarray] = []
        self._flat: Optional[torch.Tensor] = None

    def add(self, name: str, clip: np.ndarray) -> int:
        if name in self.names:
            return self.names[name]
        clip = np.ascontiguousarray(clip, dtype=np.float32).reshape(-1)
        sid = len(self.lengths)
        self.names[name] = sid
        self.offsets.append(sum(self.lengths))
        self.lengths.append(int(clip.shape[0]))
        self._host.append(clip)
        self._flat = None
        return sid

    def flat(self) -> torch.Tensor:
        if self._flat is None:
            self._flat = torch.from_numpy(np.concatenate(self._host)).to(self.device)
        return self._flat

    def __len__(self):
        return len(self.lengths)


class RirBank:
    """Binaural RIRs, planar float32 ``[R, 2, cap]`` on the device, rows zero beyond their own length
    (the precondition of ss_fftconv_binaural_f32)."""

    def __init__(self, data: torch.Tensor, lengths: torch.Tensor):
        assert data.dim() == 3 and data.shape[1] == 2 and data.dtype == torch.float32 and data.is_contiguous()
        assert lengths.dtype == torch.int32 and lengths.shape == (data.shape[0],)
        self.data, self.lengths = data, lengths
        self.cap = int(data.shape[2])

    @staticmethod
    def from_arrays(rirs: Sequence[Optional[np.ndarray]], device, cap: Optional[int] = None) -> "RirBank":
        """rirs[i]: float array [L, 2] (wav layout, what scipy.io.wavfile.read returns) or [2, L];
        None or an empty array = unreadable / empty file -> zero RIR (simulator.py:619-624)."""
        norm = []
        for r in rirs:
            if r is None or np.size(r) == 0:
                norm.append(np.zeros((2, 0), np.float32))
                continue
            r = np.asarray(r, dtype=np.float32)
            norm.append(np.ascontiguousarray(r.T if (r.ndim == 2 and r.shape[1] == 2 and r.shape[0] != 2) else r))
        longest = max([a.shape[1] for a in norm] + [2])
        cap = cap or longest
        cap += cap & 1                       # even capacity: 8-byte aligned rows for the float2 loads
        assert cap >= longest
        host = np.zeros((len(norm), 2, cap), np.float32)
        for i, a in enumerate(norm):
            host[i, :, :a.shape[1]] = a
        lens = np.array([a.shape[1] for a in norm], np.int32)
        return RirBank(torch.from_numpy(host).to(device), torch.from_numpy(lens).to(device))

    def __len__(self):
        return int(self.data.shape[0])


@dataclass
class UnitRequest:
    """What one env contributes per step (the reference state read by _compute_audiogoal):
    sound / t0 from (_current_sound, _audio_index) or (_current_sample_index); rir = bank slot of
    (scene, azimuth, receiver, source); silent = _episode_step_count > _duration."""
    sound: int = 0
    t0: int = 0
    rir: int = -1
    silent: bool = False
    dis_sound: int = -1
    dis_rir: int = -1


@dataclass
class Plan:
    """Device-side unit descriptors of one batch + what the host knows about them."""
    desc: torch.Tensor            # int32 [N, 8]
    flags: int = 0                # ops.FLAG_* promises (e.g. no unit carries a distractor term)

    def __len__(self):
        return int(self.desc.shape[0])


class BatchedAudioRenderer:
    def __init__(self, sampling_rate: int, device="cuda", pad_mode: str = "reflect",
                 step_time: Optional[float] = None, wrap: bool = False, spec_capacity: int = 64):
        """step_time None -> SoundSpaces 1.0 semantics (1-s observations).  step_time = 0.25 with wrap=True ->
        SoundSpaces 2.0 (_convolve_with_rir): int(sr*step_time) samples computed, zero-padded to 1 s."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ops._lib.SsHipError("BatchedAudioRenderer needs an MI355X (device='cuda'); there is no CPU path")
        ops.init()
        self.sr = int(sampling_rate)
        self.out_len = self.sr
        self.n_valid = self.sr if step_time is None else int(self.sr * step_time)
        self.wrap = bool(wrap)
        self.pad_mode = pad_mode
        self.nby = max(1, P.ceil_div(self.n_valid, P.KB))
        self.sources = SourceBank(self.device)
        self.rirs: Optional[RirBank] = None
        self._spec = torch.empty((spec_capacity, P.SPEC_FLOATS), dtype=torch.float32, device=self.device)
        self._n_slots = 0
        self._windows: Dict[Tuple[int, int, bool], Tuple[int, P.WindowSet]] = {}
        self.spectrogram_shape = P.spectrogram_shape(self.out_len)

    # ---- banks ---------------------------------------------------------------------------------------
    def add_source(self, name: str, clip: np.ndarray) -> int:
        return self.sources.add(name, clip)

    def set_rir_bank(self, bank: RirBank) -> None:
        if self.rirs is not None and bank.cap != self.rirs.cap:
            self.clear_window_cache()            # the set of needed partition offsets depends on the capacity
        self.rirs = bank

    def clear_window_cache(self) -> None:
        self._windows.clear()
        self._n_slots = 0

    # ---- planning --------------------------------------------------------------------------------------
    def _ensure_windows(self, keys) -> None:
        new = [k for k in dict.fromkeys(keys) if k not in self._windows]
        if not new:
            return
        nbh_max = max(1, P.ceil_div(self.rirs.cap, P.KB))
        rows, first = [], self._n_slots
        for (sid, t0, wrap) in new:
            ws = P.plan_window_set(self.sources.lengths[sid], t0, nbh_max, self.nby, wrap)
            self._windows[(sid, t0, wrap)] = (self._n_slots, ws)
            rows.append(P.window_desc_rows(ws, self.sources.offsets[sid], self.sources.lengths[sid], wrap))
            self._n_slots += ws.count
        if self._n_slots > self._spec.shape[0]:
            grown = torch.empty((max(self._n_slots, 2 * self._spec.shape[0]), P.SPEC_FLOATS),
                                dtype=torch.float32, device=self.device)
            grown[:first] = self._spec[:first]
            self._spec = grown
        wd = np.concatenate(rows) if rows else np.zeros((0, 4), np.int32)
        if len(wd):
            wd_dev = torch.from_numpy(np.ascontiguousarray(wd)).to(self.device)
            ops.source_windows_into(self.sources.flat(), wd_dev, self._spec[first:first + len(wd)])

    def plan(self, units: Sequence[UnitRequest]) -> Plan:
        """-> unit descriptors (int32 [N, 8]) on the device; computes any missing source-window spectra."""
        assert self.rirs is not None, "set_rir_bank() first"
        keys = []
        for u in units:
            if not u.silent and u.rir >= 0:
                keys.append((u.sound, u.t0, self.wrap))
                if u.dis_rir >= 0:
                    keys.append((u.dis_sound, 0, False))          # distractor: whole clip, full conv (:659-664)
        self._ensure_windows(keys)
        desc = np.zeros((len(units), 8), np.int32)
        flags = ops.FLAG_NO_DISTRACTOR
        for n, u in enumerate(units):
            if u.silent or u.rir < 0:
                desc[n] = P.unit_desc_row()
                continue
            s0, ws = self._windows[(u.sound, u.t0, self.wrap)]
            if u.dis_rir >= 0:
                d0, dws = self._windows[(u.dis_sound, 0, False)]
                desc[n] = P.unit_desc_row(u.rir, s0, ws, u.dis_rir, d0, dws)
                flags = 0
            else:
                desc[n] = P.unit_desc_row(u.rir, s0, ws)
        return Plan(torch.from_numpy(desc).to(self.device, non_blocking=True), flags)

    def plan_arrays(self, sound: np.ndarray, t0: np.ndarray, rir: np.ndarray) -> Plan:
        """Vectorised plan() for the common no-distractor case (rir < 0 = silent): the per-step host cost is a handful
        of numpy operations on the N-vectors plus one dict lookup per *distinct* (sound, t0) pair."""
        assert self.rirs is not None, "set_rir_bank() first"
        sound = np.asarray(sound, np.int64)
        t0 = np.asarray(t0, np.int64)
        rir = np.asarray(rir, np.int64)
        active = rir >= 0
        desc = np.zeros((sound.shape[0], 8), np.int32)
        desc[:, 0] = -1
        desc[:, 4] = -1
        if active.any():
            keys, inv = np.unique(np.stack([sound[active], t0[active]], axis=1), axis=0, return_inverse=True)
            self._ensure_windows([(int(s), int(t), self.wrap) for s, t in keys])
            tab = np.array([(self._windows[(int(s), int(t), self.wrap)][0],) +
                            (lambda ws: (ws.m_min, ws.count))(self._windows[(int(s), int(t), self.wrap)][1])
                            for s, t in keys], np.int32).reshape(-1, 3)
            rows = tab[inv.reshape(-1)]
            ok = rows[:, 2] > 0                                  # windows with nothing to convolve -> silent
            idx = np.flatnonzero(active)[ok]
            desc[idx, 0] = rir[active][ok]
            desc[idx, 1:4] = rows[ok]
        return Plan(torch.from_numpy(desc).to(self.device, non_blocking=True), ops.FLAG_NO_DISTRACTOR)

    # ---- rendering ---------------------------------------------------------------------------------------
    def render(self, plan: Plan, want_audiogoal: bool = False,
               audiogoal_out: Optional[torch.Tensor] = None, spectrogram_out: Optional[torch.Tensor] = None):
        """One launch (two for rows longer than one partition block) on the current stream.
        Returns (audiogoal [N,2,sr] or None, spectrogram [N,65,T4,2])."""
        N = len(plan)
        need_ag = want_audiogoal or audiogoal_out is not None or self.out_len > P.KB
        ag = audiogoal_out
        if need_ag and ag is None:
            ag = torch.empty((N, 2, self.out_len), dtype=torch.float32, device=self.device)
        sg = spectrogram_out
        if sg is None:
            sg = torch.empty((N,) + self.spectrogram_shape, dtype=torch.float32, device=self.device)
        ops.audio_obs_into(self._spec, self.rirs.data, self.rirs.lengths, plan.desc, ag, sg, self.n_valid,
                           self.out_len, self.pad_mode, flags=plan.flags)
        return (ag if (want_audiogoal or audiogoal_out is not None) else None), sg

    def render_audiogoal(self, plan: Plan, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """AudioGoalSensor-only configurations (soundspaces/tasks/nav.py:37-60)."""
        if out is None:
            out = torch.empty((len(plan), 2, self.out_len), dtype=torch.float32, device=self.device)
        ops.fftconv_binaural_into(self._spec, self.rirs.data, self.rirs.lengths, plan.desc, out, self.n_valid,
                                  flags=plan.flags)
        return out

    def render_crossfaded(self, desc_last: Plan, desc_cur: Plan):
        """SS2.0 CROSSFADE (continuous_simulator.py:47-53, 422-424): the step is convolved with the previous and the
        current RIR and blended by a linear ramp over int(0.05*sr)+1 samples.  Two convolution launches, a torch
        blend on the first 801/2206 samples, then the stand-alone spectrogram kernel."""
        a_last = self.render_audiogoal(desc_last)
        a_cur = self.render_audiogoal(desc_cur)
        n = int(0.05 * self.sr)
        w2 = torch.arange(n + 1, device=self.device, dtype=torch.float32) / n
        a_cur[:, :, :n + 1] = a_last[:, :, :n + 1] * w2.flip(0) + a_cur[:, :, :n + 1] * w2
        return a_cur, ops.spectrogram(a_cur, self.pad_mode)


class RirStore:
    """HBM-resident RIR bank with a fixed number of slots, filled on demand and evicted LRU.

    The reference re-reads ``<rir_dir>/<azimuth>/<recv>_<src>.wav`` from disk on every cache-missing step
    (simulator.py:615-618); here the first visit of a pose pays that read + one H2D copy and later visits read HBM.
    Live RIRs (SS2.0 / habitat_sim audio sensor: a new RIR every step) use a per-env key with ``refresh=True``.
    Rows are zeroed beyond the RIR's length (precondition of the kernel); RIRs longer than ``cap`` are truncated,
    which is exact for 1-s clips as long as cap >= sr (only h[0:sr] reaches y[0:sr])."""

    def __init__(self, slots: int, cap: int, device):
        cap += cap & 1
        self.bank = RirBank(torch.zeros((slots, 2, cap), dtype=torch.float32, device=device),
                            torch.zeros((slots,), dtype=torch.int32, device=device))
        self.slots, self.cap = slots, cap
        self._slot_of: Dict[object, int] = {}      # insertion order == LRU order (oldest first)
        self._free: List[int] = list(range(slots - 1, -1, -1))
        self.hits = self.misses = 0

    def _upload(self, slot: int, rir: Optional[np.ndarray]) -> None:
        row = np.zeros((2, self.cap), np.float32)
        n = 0
        if rir is not None and np.size(rir):
            r = np.asarray(rir, dtype=np.float32)
            r = r.T if (r.ndim == 2 and r.shape[1] == 2 and r.shape[0] != 2) else r
            n = min(r.shape[1], self.cap)
            row[:, :n] = r[:, :n]
        self.bank.data[slot].copy_(torch.from_numpy(row))
        self.bank.lengths[slot] = n

    def slot(self, key, loader, refresh: bool = False) -> int:
        """Bank slot of ``key``; ``loader()`` -> float array [L,2] / [2,L] or None is only called on a miss
        (or always with ``refresh=True``: live RIRs that change every step keep their slot)."""
        if key in self._slot_of:
            slot = self._slot_of.pop(key)
            self._slot_of[key] = slot                   # most recently used
            if refresh:
                self._upload(slot, loader())
            else:
                self.hits += 1
            return slot
        self.misses += 1
        if self._free:
            slot = self._free.pop()
        else:
            victim = next(iter(self._slot_of))
            slot = self._slot_of.pop(victim)
        self._slot_of[key] = slot
        self._upload(slot, loader())
        return slot


    def _pack(self, rir: Optional[np.ndarray], row: np.ndarray) -> int:
        row[:] = 0.0
        if rir is None or not np.size(rir):
            return 0
        r = np.asarray(rir, dtype=np.float32)
        r = r.T if (r.ndim == 2 and r.shape[1] == 2 and r.shape[0] != 2) else r
        n = min(r.shape[1], self.cap)
        row[:, :n] = r[:, :n]
        return n

    def slot_many(self, keys: Sequence, loaders: Sequence, workers: int = 8) -> List[int]:
        """Slots of many keys at once (scene load, `AudioGoalBatcher`): the misses' loaders run on a thread pool
        (file reads overlap), their rows are packed into ONE pinned staging block and reach the bank with one H2D copy
        plus one row scatter, instead of one 128 KB copy per file.  Same LRU semantics as `slot()`; a batch must fit the
        store (its slots are returned together, so none of them may evict another)."""
        from concurrent.futures import ThreadPoolExecutor
        out: List[int] = [-1] * len(keys)
        todo = []
        for i, key in enumerate(keys):
            if key in self._slot_of:
                out[i] = self.slot(key, loaders[i])
            else:
                todo.append(i)
        # duplicates inside the batch load once
        first = {}
        for i in todo:
            first.setdefault(keys[i], i)
        uniq = list(first.values())
        if len(set(keys)) > self.slots:
            raise ValueError(f"slot_many: {len(set(keys))} distinct keys do not fit a store of {self.slots} slots")
        # keys of this batch that are already resident must survive the evictions below: make them most recent
        for lo in range(0, len(uniq), self.slots):
            part = uniq[lo:lo + self.slots]
            if workers > 1 and len(part) > 1:
                with ThreadPoolExecutor(max_workers=workers) as pool:
                    rirs = list(pool.map(lambda i: loaders[i](), part))
            else:
                rirs = [loaders[i]() for i in part]
            stage = torch.zeros((len(part), 2, self.cap), dtype=torch.float32,
                                pin_memory=self.bank.data.device.type == "cuda")
            stage_np = stage.numpy()
            lens = np.zeros((len(part),), np.int32)
            slots = []
            for j, i in enumerate(part):
                lens[j] = self._pack(rirs[j], stage_np[j])
                self.misses += 1
                if self._free:
                    sl = self._free.pop()
                else:
                    victim = next(iter(self._slot_of))
                    sl = self._slot_of.pop(victim)
                self._slot_of[keys[i]] = sl
                slots.append(sl)
            dev = self.bank.data.device
            idx = torch.as_tensor(slots, dtype=torch.long, device=dev)
            self.bank.data.index_copy_(0, idx, stage.to(dev, non_blocking=True))
            self.bank.lengths.index_copy_(0, idx, torch.from_numpy(lens).to(dev))
        for i in todo:
            out[i] = self._slot_of[keys[i]]
        return out


def load_scene_rirs(store: "RirStore", scene_rir_dir: str, reader, azimuths=(0, 90, 180, 270), limit: Optional[int] = None,
                    batch: int = 256, workers: int = 8):
    """Bulk pre-load of one scene's binaural RIRs, `<scene_rir_dir>/<azimuth>/<receiver>_<source>.wav`
    (soundspaces/README.md:38-42, simulator.py:615-616), into the HBM store so that no step of an episode in this scene
    touches the disk.  Keys are the file paths the simulator adapter asks for.  Returns the number of files loaded."""
    import os
    paths = []
    for az in azimuths:
        d = os.path.join(scene_rir_dir, str(az))
        if not os.path.isdir(d):
            continue
        paths += [os.path.join(d, name) for name in sorted(os.listdir(d)) if name.endswith(".wav")]
    if limit is not None:
        paths = paths[:limit]
    batch = max(1, min(batch, store.slots))
    for lo in range(0, len(paths), batch):                      # threaded reads, one H2D copy per batch
        part = paths[lo:lo + batch]
        store.slot_many(part, [(lambda path=path: reader(path)) for path in part], workers=workers)
    return len(paths)


class AudioEngine:
    """Renderer + RIR store + source registry: what ``ss_amd.sim_audio`` talks to (one per process / GPU)."""

    def __init__(self, sampling_rate: int, device="cuda", rir_slots: int = 4096, rir_cap: Optional[int] = None,
                 **renderer_kwargs):
        self.renderer = BatchedAudioRenderer(sampling_rate, device=device, **renderer_kwargs)
        self.store = RirStore(rir_slots, rir_cap or sampling_rate, self.renderer.device)
        self.renderer.set_rir_bank(self.store.bank)

    def source_id(self, name: str, clip: np.ndarray) -> int:
        return self.renderer.add_source(name, clip)

    def rir_slot(self, key, loader, refresh: bool = False) -> int:
        return self.store.slot(key, loader, refresh)

    def observe(self, units: Sequence[UnitRequest], want_audiogoal: bool = False, want_spectrogram: bool = True,
                spectrogram_out=None, audiogoal_out=None) -> Dict[str, torch.Tensor]:
        plan = self.renderer.plan(units)
        if not want_spectrogram:
            return {"audiogoal": self.renderer.render_audiogoal(plan, out=audiogoal_out)}
        ag, sg = self.renderer.render(plan, want_audiogoal=want_audiogoal, audiogoal_out=audiogoal_out,
                                      spectrogram_out=spectrogram_out)
        out = {"spectrogram": sg}
        if ag is not None:
            out["audiogoal"] = ag
        return out

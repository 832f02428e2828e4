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

import os
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
        self.spectra: Optional[torch.Tensor] = None      # [R, 2, ceil(cap/KB), SPEC_FLOATS]: see build_spectra()

    def build_spectra(self) -> torch.Tensor:
        """Spectral form of the bank (ss_rir_spectra_f32): the forward FFT of every RIR block, done once here instead
        of once per step and unit (the reference redoes it inside every fftconvolve call, simulator.py:630).  With it
        the renderer runs k_conv_spec (no forward FFT; 2x the bytes per RIR).  For banks whose entries change after
        this call, call it again (or use RirStore(spectral=True), which keeps the two forms in step)."""
        self.spectra = ops.rir_spectra(self.data)
        return self.spectra

    @staticmethod
    def from_arrays(rirs: Sequence[Optional[np.ndarray]], device, cap: Optional[int] = None) -> "RirBank":
        """rirs[i]: float array [L, 2] (wav layout, what scipy.io.wavfile.read returns) or [2, L] (a [2, 2] array is read
        as the wav layout: a file of two samples); None or an empty array = unreadable / empty file -> zero RIR
        (simulator.py:619-624)."""
        norm = []
        for r in rirs:
            if r is None or np.size(r) == 0:
                norm.append(np.zeros((2, 0), np.float32))
                continue
            r = np.asarray(r, dtype=np.float32)
            norm.append(np.ascontiguousarray(r.T if (r.ndim == 2 and r.shape[1] == 2) else r))
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


class BucketedRirBank:
    """A RIR bank as LENGTH BUCKETS (include/ss_hip.h ``ss_rir_bucket``): bucket b is a ``RirBank`` of its
    own (rows of ``cap_b`` samples) holding the global bank indices ``[first_b, first_b + len(bucket b))``; ``lengths`` is
    ONE int32 tensor over all indices (the buckets' own ``lengths`` are views of it).  A 3-s RIR lands in a long bucket
    without lengthening - or reallocating - the rows of the short ones, and launches whose units all sit in bucket 0 keep
    the loop-free kernel."""

    def __init__(self, banks: Sequence[RirBank], lengths: torch.Tensor, firsts: Optional[Sequence[int]] = None):
        assert 1 <= len(banks) <= 4
        self.banks = list(banks)
        self.first = list(firsts) if firsts is not None else list(np.cumsum([0] + [len(b) for b in banks[:-1]]))
        assert self.first[0] == 0 and all(self.first[b + 1] >= self.first[b] + len(banks[b]) for b in range(len(banks) - 1))
        self.lengths = lengths
        self._carr = None                       # (ctypes array, spectral?) cache; refresh() after a bucket was reallocated

    @property
    def cap(self) -> int:                       # planning depth: the longest bucket
        return max(b.cap for b in self.banks)

    @property
    def spectra(self):                          # "every bucket has its spectral form"
        return all(b.spectra is not None for b in self.banks) or None

    def build_spectra(self) -> None:
        for b in self.banks:
            b.build_spectra()
        self._carr = None

    def refresh(self) -> None:
        self._carr = None

    def c_array(self, spectral: bool):
        if self._carr is None or self._carr[1] != spectral or self._carr[2] != [b.data.data_ptr() for b in self.banks]:
            self._carr = (ops.bucket_array(self.banks, self.first, spectral), spectral, [b.data.data_ptr() for b in self.banks])
        return self._carr[0]

    def bucket_of(self, index: int) -> int:
        b = 0
        while b + 1 < len(self.first) and index >= self.first[b + 1]:
            b += 1
        return b

    def __len__(self):
        return int(self.lengths.shape[0])

    @staticmethod
    def from_arrays(rirs: Sequence[Optional[np.ndarray]], device, caps: Sequence[int]) -> "BucketedRirBank":
        """rirs[i] as for ``RirBank.from_arrays``; ``caps`` = ascending bucket capacities (the last one must hold the
        longest RIR).  Entry i keeps GLOBAL index order inside its bucket; returns the bank and sets ``index_of[i]`` =
        the global bank index of input i."""
        lens = [0 if (r is None or np.size(r) == 0) else max(np.shape(r)) for r in rirs]
        caps = [c + (c & 1) for c in caps]
        which = [next(b for b, c in enumerate(caps) if L <= c) for L in lens]
        banks, firsts, index_of, first = [], [], [0] * len(rirs), 0
        for b, cap in enumerate(caps):
            members = [i for i, w in enumerate(which) if w == b]
            bank = RirBank.from_arrays([rirs[i] for i in members] or [None], device, cap=cap)
            for k, i in enumerate(members):
                index_of[i] = first + k
            banks.append(bank)
            firsts.append(first)
            first += len(bank)
        lengths = torch.cat([b.lengths for b in banks])
        off = 0
        for b in banks:                                            # the buckets' lengths become views of the global tensor
            b.lengths = lengths[off:off + len(b)]
            off += len(b)
        out = BucketedRirBank(banks, lengths, firsts)
        out.index_of = index_of
        return out


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
    # SoundSpaces 2.0 only (renderer built with wrap=True):
    # wrap: the reference wraps the clip around ONLY in its steady branch (continuous_simulator.py:438-447,
    #   index >= rir length); the early branch slices source[:index+num_sample] and reads zeros past the clip end
    #   (:433-437).  None = "steady branch" (wrap), False = early branch.
    # last_rir: bank slot of the previous step's RIR (`_last_rir`) for CROSSFADE (:422-424), -1 = none; last_wrap is
    #   the branch `_convolve_with_rir(self._last_rir)` takes for THAT RIR's length.
    wrap: Optional[bool] = None
    last_rir: int = -1
    last_wrap: Optional[bool] = None


@dataclass
class Plan:
    """Device-side unit descriptors of one batch + what the host knows about them."""
    desc: torch.Tensor            # int32 [N, 8]
    flags: int = 0                # ops.FLAG_* promises (e.g. no unit carries a distractor term)

    def __len__(self):
        return int(self.desc.shape[0])


class BatchedAudioRenderer:
    def __init__(self, sampling_rate: int, device="cuda", pad_mode: str = "reflect",
                 step_time: Optional[float] = None, wrap: bool = False, spec_capacity: int = 64,
                 max_window_slots: int = 2048):
        """step_time None -> SoundSpaces 1.0 semantics (1-s observations).  step_time = 0.25 with wrap=True ->
        SoundSpaces 2.0 (_convolve_with_rir): int(sr*step_time) samples computed, zero-padded to 1 s.
        max_window_slots bounds the cache of source-window spectra (128 KiB of HBM each): SS1.0 keys (sound, index*sr)
        are a small fixed set, but SS2.0 draws a new sample index per env and step, so the cache is emptied at the start
        of the plan() that finds it over the bound (plans made before that call must have been rendered: they are by
        every caller in this package, which plans and renders a step at a time; pre-planned batches share their keys)."""
        self.max_window_slots = int(max_window_slots)
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
        if self._n_slots > self.max_window_slots:       # bounded: start over, this step's keys are recomputed
            self._windows.clear()
            self._n_slots = 0
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

    def _wrap_of(self, sound: int, t0: int, wrap: Optional[bool]) -> bool:
        """Effective wrap flag of a window key: only SS2.0 renderers wrap, only in the reference's steady branch, and
        only windows that actually run past the clip end differ from the non-wrapping ones (fewer distinct keys)."""
        if not self.wrap or wrap is False:
            return False
        return t0 + self.n_valid > self.sources.lengths[sound]

    def plan(self, units: Sequence[UnitRequest]) -> Plan:
        """-> unit descriptors (int32 [N, 8]) on the device; computes any missing source-window spectra."""
        assert self.rirs is not None, "set_rir_bank() first"
        keys, ukeys = [], []
        for u in units:
            k0 = k1 = None
            if not u.silent and u.rir >= 0:
                k0 = (u.sound, u.t0, self._wrap_of(u.sound, u.t0, u.wrap))
                keys.append(k0)
                if u.last_rir >= 0:                               # CROSSFADE: same window, previous RIR's branch
                    assert u.dis_rir < 0, "a unit carries either a distractor or a previous RIR in term 1"
                    k1 = (u.sound, u.t0, self._wrap_of(u.sound, u.t0, u.wrap if u.last_wrap is None else u.last_wrap))
                    keys.append(k1)
                elif u.dis_rir >= 0:
                    k1 = (u.dis_sound, 0, False)                  # distractor: whole clip, full conv (:659-664)
                    keys.append(k1)
            ukeys.append((k0, k1))
        self._ensure_windows(keys)
        desc = np.zeros((len(units), 8), np.int32)
        flags = ops.FLAG_NO_DISTRACTOR
        xfade = any(u.last_rir >= 0 and not u.silent and u.rir >= 0 for u in units)
        for n, u in enumerate(units):
            k0, k1 = ukeys[n]
            if k0 is None:
                desc[n] = P.unit_desc_row()
                continue
            s0, ws = self._windows[k0]
            if k1 is not None:
                assert xfade == (u.last_rir >= 0), "cross-faded and distractor units cannot share a launch"
                d0, dws = self._windows[k1]
                desc[n] = P.unit_desc_row(u.rir, s0, ws, u.last_rir if xfade else u.dis_rir, d0, dws)
                flags = 0
            else:
                desc[n] = P.unit_desc_row(u.rir, s0, ws)
        if xfade:
            flags = ops.FLAG_CROSSFADE
        flags |= self._bucket_flag(max([max(u.rir, u.dis_rir, u.last_rir) for u in units if not u.silent] + [-1]))
        return Plan(torch.from_numpy(desc).to(self.device, non_blocking=True), flags)

    def _bucket_flag(self, max_index: int) -> int:
        """SS_FLAG_FIRST_BUCKET for a launch on a bucketed bank whose highest bank index is `max_index`."""
        if isinstance(self.rirs, BucketedRirBank) and len(self.rirs.first) > 1 and max_index < self.rirs.first[1]:
            return ops.FLAG_FIRST_BUCKET
        return 0

    def plan_arrays(self, sound: np.ndarray, t0: np.ndarray, rir: np.ndarray, rotations: int = 1) -> Plan:
        """Vectorised plan() for the common no-distractor case (rir < 0 = silent): the per-step host cost is a handful
        of numpy operations on the N-vectors plus one dict lookup per *distinct* (sound, t0) pair.
        ``rotations`` = R > 1: every env yields R units (unit n*R + k) that hear the same clip window through the R
        ADJACENT bank rows rir[n] + k -- the azimuths of one (receiver, source) pair stored side by side
        (RirStore(group=R); BASELINE configs[2]: "4 agent rotations per step"), so the 2R rows of an env are one
        contiguous stretch of the bank."""
        assert self.rirs is not None, "set_rir_bank() first"
        sound = np.asarray(sound, np.int64)
        t0 = np.asarray(t0, np.int64)
        rir = np.asarray(rir, np.int64)
        if rotations > 1:
            k = np.arange(rotations, dtype=np.int64)
            sound, t0 = np.repeat(sound, rotations), np.repeat(t0, rotations)
            rir = np.where(rir[:, None] >= 0, rir[:, None] + k[None, :], -1).reshape(-1)
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
        return Plan(torch.from_numpy(desc).to(self.device, non_blocking=True),
                    ops.FLAG_NO_DISTRACTOR | self._bucket_flag(int(rir.max()) if rir.size else -1))

    # ---- rendering ---------------------------------------------------------------------------------------
    # With BOTH bank forms resident, launches of more than this many units of one-block rows (16 kHz) read the time-domain rows
    # (ss_ctx_set_spectral_policy has the numbers); 0 = the spectral form whenever it exists.  AudioEngine sets it.
    spectral_max_units = 0

    def _spectral_for(self, n_units: int, distractor: bool = False) -> bool:
        """(steps with distractor terms - two forward transforms per row - read the spectral rows at any size)"""
        return self.spectral_max_units <= 0 or self.out_len > P.KB or n_units <= self.spectral_max_units or distractor

    def render(self, plan: Plan, want_audiogoal: bool = False,
               audiogoal_out: Optional[torch.Tensor] = None, spectrogram_out: Optional[torch.Tensor] = None):
        """One launch on the current stream - including SS2.0 steps at 44.1 kHz (one rendered block of a longer row: the
        fused loop kernel, k_conv<..., WIDE>); two - convolution, then spectrogram, the waveform handed over through memory -
        for cross-faded rows with more than one rendered block (measured faster than k_obs_rows<XFADE>) and for rows longer
        than three blocks.  Returns (audiogoal [N,2,sr] or None, spectrogram)."""
        N = len(plan)
        xfade = bool(plan.flags & ops.FLAG_CROSSFADE)
        spectral = (self.rirs.spectra is not None if not isinstance(self.rirs, BucketedRirBank) else bool(self.rirs.spectra)) \
            and not xfade and self._spectral_for(N, not (plan.flags & ops.FLAG_NO_DISTRACTOR))
        need_ag = (want_audiogoal or audiogoal_out is not None or
                   (self.out_len > P.KB and not P.wide_one_block(self.out_len, self.n_valid, spectral)
                    and (xfade or self.n_valid <= P.KB or self.out_len > 3 * P.KB)))
        ag = audiogoal_out
        if need_ag and ag is None:
            ag = torch.empty((N, 2, self.out_len), dtype=torch.float32, device=self.device)
        sg = spectrogram_out
        if sg is None:
            sg = torch.empty((N,) + self.spectrogram_shape, dtype=torch.float32, device=self.device)
        if isinstance(self.rirs, BucketedRirBank):
            ops.audio_obs_buckets_into(self._spec, self.rirs.c_array(spectral), len(self.rirs.banks), self.rirs.lengths,
                                       plan.desc, ag, sg, self.n_valid, self.out_len, self.pad_mode, flags=plan.flags)
        elif spectral:
            ops.audio_obs_spec_into(self._spec, self.rirs.spectra, self.rirs.lengths, plan.desc, ag, sg, self.n_valid,
                                    self.out_len, self.pad_mode, flags=plan.flags)
        else:
            ops.audio_obs_into(self._spec, self.rirs.data, self.rirs.lengths, plan.desc, ag, sg, self.n_valid,
                               self.out_len, self.pad_mode, flags=plan.flags)
        return (ag if (want_audiogoal or audiogoal_out is not None) else None), sg

    def render_audiogoal(self, plan: Plan, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """AudioGoalSensor-only configurations (soundspaces/tasks/nav.py:37-60)."""
        if out is None:
            out = torch.empty((len(plan), 2, self.out_len), dtype=torch.float32, device=self.device)
        if isinstance(self.rirs, BucketedRirBank):
            spectral = bool(self.rirs.spectra) and not (plan.flags & ops.FLAG_CROSSFADE) and \
                self._spectral_for(len(plan), not (plan.flags & ops.FLAG_NO_DISTRACTOR))
            ops.audio_obs_buckets_into(self._spec, self.rirs.c_array(spectral), len(self.rirs.banks), self.rirs.lengths,
                                       plan.desc, out, None, self.n_valid, self.out_len, self.pad_mode, flags=plan.flags)
        elif self.rirs.spectra is not None and not (plan.flags & ops.FLAG_CROSSFADE) and \
                self._spectral_for(len(plan), not (plan.flags & ops.FLAG_NO_DISTRACTOR)):
            ops.fftconv_binaural_spec_into(self._spec, self.rirs.spectra, self.rirs.lengths, plan.desc, out, self.n_valid,
                                           flags=plan.flags)
        else:
            ops.fftconv_binaural_into(self._spec, self.rirs.data, self.rirs.lengths, plan.desc, out, self.n_valid,
                                      flags=plan.flags)
        return out

    def render_crossfaded(self, units: Sequence[UnitRequest], want_audiogoal: bool = True):
        """SS2.0 CROSSFADE (continuous_simulator.py:47-53, 422-424) in ONE launch: every unit carries the previous
        step's RIR in ``last_rir``; the loop kernel convolves the step with it first, keeps the ramp's
        int(0.05*sr)+1 samples on the CU, convolves with the current RIR and blends in registers; the spectrogram is
        taken from the blended row (fused at 16 kHz).  Returns (audiogoal or None, spectrogram)."""
        return self.render(self.plan(units), want_audiogoal=want_audiogoal)


def _planar(rir) -> np.ndarray:
    """float32 [2, L] from a wav-layout [L, 2] / planar [2, L] array ([2, 2] = wav layout); None / empty -> [2, 0]."""
    if rir is None or not np.size(rir):
        return np.zeros((2, 0), np.float32)
    r = np.asarray(rir, dtype=np.float32)
    return r.T if (r.ndim == 2 and r.shape[1] == 2) else r


class RirStore:
    """HBM-resident RIR bank with a fixed number of slots, filled on demand and evicted LRU.

    The reference re-reads ``<rir_dir>/<azimuth>/<recv>_<src>.wav`` from disk on every cache-missing step
    (simulator.py:615-618); here the first visit of a pose pays that read + one H2D copy and later visits read HBM.
    Live RIRs (SS2.0 / habitat_sim audio sensor: a new RIR every step) use a per-env key with ``refresh=True``.
    Rows are zeroed beyond the RIR's length (precondition of the kernel).

    Row capacity.  ``truncate_to = n`` keeps only h[0:n] of every RIR: exact as long as every clip is a 1-s clip
    convolved from its start (simulator.py:629-632: only h[0:sr] reaches y[0:sr]) and keeps rows <= one partition
    block (the loop-free kernel).  With ``truncate_to = None`` (multi-second sounds, simulator.py:634-647, and SS2.0,
    whose outputs depend on the whole RIR) nothing is ever cut: a longer RIR GROWS the bank (reallocated at the new
    capacity, contents kept, ``on_grow(bank)`` tells the renderer), up to ``max_cap`` samples, beyond which it raises.

    ``group`` > 1 stores the entries of one key in ``group`` ADJACENT slots (the 4 azimuths of one (receiver, source)
    pair: SURVEY 8(f)2), loaded and evicted together; ``slot()`` then returns the first slot of the group."""

    def __init__(self, slots: int, cap: int, device, truncate_to: Optional[int] = None, max_cap: int = 1 << 18,
                 on_grow=None, group: int = 1, spectral: bool = False):
        cap += cap & 1
        assert slots % group == 0
        self.device = torch.device(device)
        self.bank = RirBank(torch.zeros((slots, 2, cap), dtype=torch.float32, device=self.device),
                            torch.zeros((slots,), dtype=torch.int32, device=self.device))
        self.slots, self.cap, self.group = slots, cap, group
        self.truncate_to, self.max_cap, self.on_grow = truncate_to, max_cap, on_grow
        # called before ANYTHING this store writes to the bank on the device (rows, lengths, block spectra, a reallocation): the
        # engine orders the write behind its context's overlap lanes there (a step in flight may still read the entry)
        self.before_device_write = None
        self.on_evict = None                                   # on_evict(key, slot): the entry of `key` is about to be reused
        self._evict_hooks: List = []                           # ... and the same for any number of listeners (add_evict_hook)
        self.defer_uploads = False                             # True (AudioEngine): single-row uploads queue up for flush_uploads()
        self._pending: Dict[int, tuple] = {}
        self._flush_stage = None
        self._flush_stage_wav = None
        self._flush_ev = None
        self._idx_cache: Dict[tuple, torch.Tensor] = {}
        self._dev_len = np.full((slots,), -1, np.int32)       # lengths the DEVICE holds for rows written by flush_uploads
        self.host_len = np.zeros((slots,), np.int32)          # host mirror of bank.lengths (branch selection, planning)
        self._slot_of: Dict[object, int] = {}      # insertion order == LRU order (oldest first)
        self._free: List[int] = list(range(slots - group, -1, -group))
        self._free_ver = 0                                     # bumped whenever _free changes (miss_loader's mirror of its top)
        self._batch = 0
        self._batch_of = np.full((slots,), -1, np.int64)      # batch in which the slot was last handed out
        # victim selection without walking the dict (a FULL store is the steady state against an 867-GB data set: every miss
        # evicts): per first slot of an entry its key, whether it is occupied, and the tick of its last use through slot()
        # (the dict order as numbers: the tie-break among entries of one batch)
        self._key_at: List = [None] * slots
        self._used = np.zeros((slots,), bool)
        self._use_seq = np.zeros((slots,), np.int64)
        self._seq = 0
        self.hits = self.misses = self.grown = 0
        self._clipped = np.zeros((slots,), bool)              # the stored row is shorter than its RIR (truncate_to)
        # spectral=True keeps the block spectra of every row next to it (RirBank.spectra, ss_rir_spectra_f32): rows
        # (re)loaded since the last sync_spectra() are transformed there, once, instead of once per step and unit
        self.spectral = spectral
        self._stale = np.zeros((slots,), bool)
        if spectral and self.device.type == "cuda":
            self.bank.spectra = torch.zeros((slots, 2, P.ceil_div(cap, P.KB), P.SPEC_FLOATS), dtype=torch.float32,
                                            device=self.device)
        # pinned staging ring for single-row uploads (a pageable torch copy blocks the host for the whole transfer)
        self._stage = None
        self._stage_ev: List = []
        self._stage_k = 0

    def add_evict_hook(self, method) -> None:
        """Call ``method(key, slot)`` (a BOUND method, held weakly: a listener that is garbage-collected drops out) whenever an
        entry is about to be reused - every resolver that keeps a table of store slots registers here."""
        import weakref
        self._evict_hooks.append(weakref.WeakMethod(method))

    def _notify_evict(self, key, slot) -> None:
        if self.on_evict is not None:
            self.on_evict(key, slot)
        if self._evict_hooks:
            alive = []
            for ref in self._evict_hooks:
                fn = ref()
                if fn is not None:
                    fn(key, slot)
                    alive.append(ref)
            self._evict_hooks = alive

    # ---- batches -----------------------------------------------------------------------------------------
    def begin_batch(self) -> None:
        """Slots handed out from here on belong to one launch: none of them may be evicted for another key of the same
        batch (a store smaller than one step's distinct poses would otherwise render an env with another pose's RIR)."""
        self._batch += 1

    def clear(self) -> None:
        """Forget every entry (the rows are rewritten on the next miss)."""
        if self.on_evict is not None or self._evict_hooks:
            for key, slot in list(self._slot_of.items()):
                self._notify_evict(key, slot)
        self._slot_of.clear()
        self._key_at = [None] * self.slots
        self._used[:] = False
        self._pending = {}
        self._free = list(range(self.slots - self.group, -1, -self.group))
        self._free_ver += 1
        self.host_len[:] = 0
        self.bank.lengths.zero_()
        self._dev_len[:] = -1
        self._clipped[:] = False

    # ---- capacity ------------------------------------------------------------------------------------------
    def _kept_len(self, n: int) -> int:
        return n if self.truncate_to is None else min(n, self.truncate_to)

    def _about_to_write(self) -> None:
        if self.before_device_write is not None:
            self.before_device_write()

    def _ensure_cap(self, n: int) -> None:
        """Make rows at least n samples long (n = what will be stored, i.e. already clipped by truncate_to)."""
        if n <= self.cap:
            return
        if n > self.max_cap:
            raise ValueError(f"RIR of {n} samples exceeds RirStore.max_cap = {self.max_cap}")
        self._about_to_write()
        new_cap = min(self.max_cap + (self.max_cap & 1), -(-n // 2048) * 2048)
        data = torch.zeros((self.slots, 2, new_cap), dtype=torch.float32, device=self.device)
        data[:, :, :self.cap] = self.bank.data
        self.bank = RirBank(data, self.bank.lengths)
        if self.spectral and self.device.type == "cuda":        # more blocks per row: every spectrum is rebuilt
            self.bank.spectra = torch.zeros((self.slots, 2, P.ceil_div(new_cap, P.KB), P.SPEC_FLOATS),
                                            dtype=torch.float32, device=self.device)
            self._stale[:] = self.host_len > 0
        self.cap = new_cap
        self._stage = None
        self.grown += 1
        if self.on_grow is not None:
            self.on_grow(self.bank)

    # ---- uploads ---------------------------------------------------------------------------------------------
    def _stage_row(self):
        """-> (pinned [2, cap] staging row as torch + numpy views, its slot in the ring).  CPU stores: plain memory."""
        if self.device.type != "cuda":
            t = torch.zeros((2, self.cap), dtype=torch.float32)
            return t, t.numpy(), -1
        if self._stage is None:
            self._stage = torch.zeros((8, 2, self.cap), dtype=torch.float32, pin_memory=True)
            self._stage_ev = [None] * 8
        k = self._stage_k % 8
        self._stage_k += 1
        if self._stage_ev[k] is not None:
            self._stage_ev[k].synchronize()                     # the copy that last read this staging row has run
        return self._stage[k], self._stage[k].numpy(), k

    def _upload(self, slot: int, rir) -> None:
        r = _planar(rir)
        n = self._kept_len(r.shape[1])
        self._clipped[slot] = n < r.shape[1]
        self._ensure_cap(n)
        if self.defer_uploads:
            # batched mode (AudioEngine): the row crosses PCIe with the step's other new rows, as ONE pinned block and ONE
            # H2D copy, when the engine flushes before its launch (flush_uploads).  SoundSpaces 2.0 hands every env a new
            # RIR every step (continuous_simulator.py:419): 128 single-row copies + 128 one-element length fills per step
            # were the whole cost of its batched and deferred modes
            self._pending[slot] = (r, n)
            self.host_len[slot] = n
            self._stale[slot] = True
            return
        row_t, row, k = self._stage_row()
        row[:, :n] = r[:, :n]
        row[:, n:] = 0.0
        self._about_to_write()
        self.bank.data[slot].copy_(row_t, non_blocking=True)
        if k >= 0:
            ev = torch.cuda.Event()
            ev.record()
            self._stage_ev[k] = ev
        self.bank.lengths[slot:slot + 1].fill_(n)
        self._dev_len[slot] = n
        self.host_len[slot] = n
        self._stale[slot] = True

    def flush_uploads(self) -> int:
        """defer_uploads mode: every row queued by slot() / _upload since the last flush -> pinned staging blocks, one H2D
        copy per block, one row scatter, one length scatter.  Rows that arrived in wav layout ([L, 2] contiguous: what
        ``np.transpose`` of the ray tracer's [2][L] lists pickles to) are staged AS THEY ARE - one contiguous memcpy each -
        and transposed into the planar bank rows on the device; a strided host-side transpose of 128 x
        72 KB per step was most of the trainer half of SoundSpaces 2.0's deferred mode.  Returns the rows uploaded."""
        if not self._pending:
            return 0
        self._about_to_write()
        slots = sorted(self._pending)
        rows = [self._pending[sl] for sl in slots]
        self._pending = {}
        pin = self.device.type == "cuda"
        if pin and self._flush_ev is not None:
            self._flush_ev.synchronize()                       # the copies that last read the staging blocks have run
        wav = [j for j, (r, n) in enumerate(rows) if n > 0 and r.T.flags.c_contiguous and not r.flags.c_contiguous]
        wav_set = set(wav)
        pla = [j for j in range(len(rows)) if j not in wav_set]
        lens = np.asarray([n for _, n in rows], np.int32)

        def stage(which, shape_tail, attr):
            blk = getattr(self, attr)
            if blk is None or blk.shape[0] < len(which) or tuple(blk.shape[1:]) != shape_tail:
                blk = torch.zeros((max(len(which), 64),) + shape_tail, dtype=torch.float32, pin_memory=pin)
                setattr(self, attr, blk)
            return blk, blk.numpy()

        def fill(jobs):                                        # (a thread pool for these ~5-us copies was 2.6x SLOWER: GIL hand-offs)
            for f in jobs:
                f()
        jobs = []
        if pla:
            pblk, pnp = stage(pla, (2, self.cap), "_flush_stage")
            for i, j in enumerate(pla):
                r, n = rows[j]

                def job(i=i, r=r, n=n):
                    pnp[i, :, :n] = r[:, :n]
                    pnp[i, :, n:] = 0.0
                jobs.append(job)
        if wav:
            wblk, wnp = stage(wav, (self.cap, 2), "_flush_stage_wav")
            for i, j in enumerate(wav):
                r, n = rows[j]

                def job(i=i, r=r, n=n):
                    wnp[i, :n, :] = r.T[:n]                       # contiguous [n, 2] -> contiguous [n, 2]
                    wnp[i, n:, :] = 0.0
                jobs.append(job)
        fill(jobs)
        for which, blk_name, transpose in ((pla, "_flush_stage", False), (wav, "_flush_stage_wav", True)):
            if not which:
                continue
            k = len(which)
            sl = [slots[j] for j in which]
            dev_blk = getattr(self, blk_name)[:k].to(self.device, non_blocking=True)
            if transpose:
                dev_blk = dev_blk.permute(0, 2, 1)                 # [k, cap, 2] -> [k, 2, cap]: transposed by the scatter kernel
            # live rows alternate between two slot patterns: the index tensors are kept (a list -> device tensor conversion is a
            # synchronous pageable copy, 0.2 ms), and the length scatter is skipped while the lengths on the device are current
            key = tuple(sl)
            idx = self._idx_cache.get(key)
            if idx is None:
                if len(self._idx_cache) > 16:
                    self._idx_cache.clear()
                idx = self._idx_cache[key] = torch.as_tensor(sl, dtype=torch.long, device=self.device)
            self.bank.data.index_copy_(0, idx, dev_blk)
            sl_np = np.asarray(sl)
            if not np.array_equal(self._dev_len[sl_np], lens[which]):
                self.bank.lengths.index_copy_(0, idx, torch.from_numpy(lens[which]).to(self.device))
                self._dev_len[sl_np] = lens[which]
        if pin:
            self._flush_ev = torch.cuda.Event()
            self._flush_ev.record()
        return len(slots)

    gather_threads = 8            # host threads of upload_rows' gather (ss_rows_gather_f32)

    def upload_rows(self, slots: Sequence[int], rows: Sequence[np.ndarray], threads: int = 0) -> None:
        """rows[i] (float32, wav layout [L, 2], C-contiguous: what the ray tracer's output transposes to) -> bank row
        slots[i], NOW: the rows are gathered into one pinned block by the library (ss_rows_gather_f32: plain threads, no
        per-row numpy call), cross PCIe as one copy and are transposed into the planar rows by the scatter on the device.
        The vectorised form of ``slot(key, loader, refresh=True)`` + ``flush_uploads()`` for callers that own their slots (the
        live RIRs of a SoundSpaces 2.0 step: ``DeferredResolver``); rows in any other layout take ``_upload``."""
        from . import _lib
        import ctypes
        k = len(slots)
        if k == 0:
            return
        self._about_to_write()
        ok = all(r.strides == (8, 4) and r.dtype.char == "f" for r in rows)      # float32 [L, 2], C-contiguous
        if not ok:
            for sl, r in zip(slots, rows):
                self._upload(int(sl), r)
            return
        full = np.fromiter((r.shape[0] for r in rows), np.int64, k)
        lens = full if self.truncate_to is None else np.minimum(full, self.truncate_to)
        self._ensure_cap(int(lens.max()))
        if self._pending:
            for sl in slots:
                self._pending.pop(int(sl), None)
        pin = self.device.type == "cuda"
        if pin and self._flush_ev is not None:
            self._flush_ev.synchronize()                       # the copy that last read the staging block has run
        blk = self._flush_stage_wav
        if blk is None or blk.shape[0] < k or tuple(blk.shape[1:]) != (self.cap, 2):
            blk = self._flush_stage_wav = torch.zeros((max(k, 64), self.cap, 2), dtype=torch.float32, pin_memory=pin)
        # the rows' addresses: `__array_interface__` builds a dict per array (1.2 us each: 150 us of a 128-env SS2.0 step);
        # a ctypes view of the buffer gives the address in a quarter of that (read-only / empty arrays: the slow way)
        fb, ao = ctypes.c_char.from_buffer, ctypes.addressof
        try:
            addrs = [ao(fb(r)) for r in rows]
        except (TypeError, ValueError):
            addrs = [r.__array_interface__["data"][0] for r in rows]
        ptrs = np.array(addrs, np.uint64)
        nfl = (2 * lens).astype(np.int32)
        _lib.check(_lib.load().ss_rows_gather_f32(ptrs.ctypes.data, nfl.ctypes.data, k, blk.data_ptr(),
                                                  2 * self.cap, 2 * self.cap, threads if threads > 0 else self.gather_threads),
                   "ss_rows_gather_f32")
        sl_np = np.asarray(slots, np.int64)
        lens32 = lens.astype(np.int32)
        meta = getattr(self, "_flush_meta", None)
        if meta is None or meta[0].shape[0] < k:
            meta = self._flush_meta = tuple(torch.zeros((max(k, 64),), dtype=torch.int32, pin_memory=pin) for _ in range(2))
        meta[0].numpy()[:k] = sl_np
        meta[1].numpy()[:k] = lens32
        self._flush_ev = self._scatter_staged(blk, meta[0], meta[1], k)
        self._dev_len[sl_np] = lens32
        self.host_len[sl_np] = lens32
        self._clipped[sl_np] = lens < full
        self._stale[sl_np] = True

    def sync_spectra(self) -> int:
        """spectral stores: transform the rows loaded since the last call (contiguous runs, one ss_rir_spectra_f32 each;
        synchronous - this is bank-load work, steady-state steps find nothing to do).  Returns the rows transformed."""
        self.flush_uploads()
        if not self.spectral or self.bank.spectra is None or not self._stale.any():
            return 0
        self._about_to_write()
        idx = np.flatnonzero(self._stale)
        run_start = prev = int(idx[0])
        for i in list(idx[1:]) + [None]:
            if i is not None and int(i) == prev + 1:
                prev = int(i)
                continue
            ops.rir_spectra_into(self.bank.data, self.bank.spectra, run_start, prev - run_start + 1)
            if i is not None:
                run_start = prev = int(i)
        self._stale[:] = False
        return int(idx.shape[0])

    def _bind(self, key, slot: int) -> None:
        """`key` now owns the entry starting at `slot` (handed out for the current batch, most recently used)"""
        self._slot_of[key] = slot
        self._key_at[slot] = key
        self._used[slot] = True
        self._seq += 1
        self._use_seq[slot] = self._seq
        self._batch_of[slot] = self._batch

    def release(self, key) -> None:
        """Give the entry of `key` back (no eviction notice: the caller moves the key elsewhere)."""
        slot = self._slot_of.pop(key)
        self._key_at[slot] = None
        self._used[slot] = False
        self._free.append(slot)
        self._free_ver += 1

    def _take_slot(self) -> int:
        return self._take_slots(1)[0]

    def _take_slots(self, k: int) -> List[int]:
        """k entries for new keys: free ones first, then the least recently used - evicted (hooks told) in LRU order.
        Victim = the entry whose slot was handed out / touched longest ago.  Recency lives in `_batch_of` (the batch of
        the last use: slot() and the column paths' touch_slots() both write it; the native record path,
        ss_ctx_observe_requests, stamps it from C), NOT only in the dict order - with the dict order alone the store degraded
        to FIFO for exactly the modes that keep the most poses resident.  Ties (callers that never open a batch,
        entries of one batch) fall back to the order of use through slot(): oldest first.  All numpy over the slot arrays:
        the first version rebuilt a list of the dict per victim (150 us per miss at 4096 resident poses)."""
        out: List[int] = []
        self._free_ver += 1
        while self._free and len(out) < k:
            out.append(self._free.pop())
        r = k - len(out)
        if r <= 0:
            return out
        cand = np.flatnonzero(self._used)
        full_msg = (f"RirStore: {self.slots // self.group} entries cannot hold the distinct RIRs of one batch "
                    "(an entry handed out for this launch would be overwritten); raise rir_slots")
        if cand.shape[0] < r:
            self._free.extend(reversed(out))
            raise RuntimeError(full_msg)
        last = self._batch_of[cand]
        thr = np.partition(last, r - 1)[r - 1] if r < cand.shape[0] else last.max()
        if self._batch and thr == self._batch:                       # (callers that never open a batch: no guard)
            self._free.extend(reversed(out))
            raise RuntimeError(full_msg)
        below, ties = cand[last < thr], cand[last == thr]
        need = r - below.shape[0]
        if need < ties.shape[0]:
            ties = ties[np.argpartition(self._use_seq[ties], need - 1)[:need]]
        victims = np.concatenate([below, ties])
        victims = victims[np.lexsort((self._use_seq[victims], self._batch_of[victims]))]
        tell = self.on_evict is not None or bool(self._evict_hooks)
        for slot in victims.tolist():
            victim = self._key_at[slot]
            del self._slot_of[victim]
            self._key_at[slot] = None
            self._used[slot] = False
            for g in range(self.group):                              # a row still queued for the victim must not land later
                self._pending.pop(slot + g, None)
            if tell:
                self._notify_evict(victim, slot)
            out.append(slot)
        return out

    # ---- the miss path inside the library (ss_ctx_observe_requests_load) ---------------------------------------------------
    _LOADER_FREE = 64                                            # free entries lent per call (= new poses one call may load)

    def miss_loader(self, table_dirs: Sequence[str], pair_keys: np.ndarray, pair_slots: np.ndarray, n_pairs: int, threads: int = 0):
        """struct ss_miss_loader over THIS store (GPU stores, group == 1): the library may then serve a step's pose misses itself -
        files read by its reader into a pinned block of this store, one scatter launch into free entries, `pair_keys` /
        `pair_slots` (int64 buffers with spare capacity, `n_pairs` entries used) extended in place.  The returned dict is reused
        from step to step (``refresh_loader`` before a call, ``adopt_loaded`` after one that loaded something)."""
        import ctypes
        from . import _lib
        assert self.device.type == "cuda" and self.group == 1
        L = _lib.SsMissLoader()
        pin = dict(dtype=torch.int32, pin_memory=True)
        rows = self._LOADER_FREE
        d = dict(s=L, ref=ctypes.byref(L), free=np.zeros((rows,), np.int32), free_ver=-1, free_n=0,
                 loaded_key=np.zeros((rows,), np.int64), loaded_slot=np.zeros((rows,), np.int32), loaded_frames=np.zeros((rows,), np.int32),
                 evicted=np.zeros((rows,), np.int32),
                 stage_slot=torch.zeros((rows,), **pin), stage_len=torch.zeros((rows,), **pin), stage=None, cap=-1, bank=None,
                 dirs=None, dirs_arr=None, pk=None)
        L.free_slots = d["free"].ctypes.data
        L.loaded_key, L.loaded_slot, L.loaded_frames = (d[k_].ctypes.data for k_ in ("loaded_key", "loaded_slot", "loaded_frames"))
        L.loaded_cap = L.stage_rows = L.evict_cap = rows
        L.evicted_slot = d["evicted"].ctypes.data
        L.stage_slot, L.stage_len = d["stage_slot"].data_ptr(), d["stage_len"].data_ptr()
        L.threads = int(threads) if threads > 0 else min(8, os.cpu_count() or 1)
        self.refresh_loader(d, table_dirs, pair_keys, pair_slots)
        return d

    def refresh_loader(self, d, table_dirs: Sequence[str], pair_keys: np.ndarray, pair_slots: np.ndarray) -> None:
        """Point the loader at what may have been replaced since its last use (bank after a growth, pair buffers after a merge,
        a new RIR directory) and refill its stack of free entries when the store's free list changed.  Cheap when nothing did."""
        import ctypes
        L = d["s"]
        bank = self.bank
        if (d["bank"] is bank.data and d["cap"] == self.cap and d["free_ver"] == self._free_ver and L.n_free == d["free_n"] and
                d["dirs"] is table_dirs and L.n_table_dirs == len(table_dirs) and d["pk"] is pair_keys and
                d.get("keep") == self.truncate_to and not (L.n_free < 16 and len(self._free) > L.n_free)):
            return                                               # nothing changed since the last call (the steady state)
        d["keep"] = self.truncate_to
        if d["bank"] is not bank.data or d["cap"] != self.cap:
            if d["stage"] is None or d["cap"] != self.cap:
                if d["stage"] is not None:
                    torch.cuda.synchronize(self.device)          # (a scatter may still be reading the old block)
                d["stage"] = torch.zeros((L.stage_rows, self.cap, 2), dtype=torch.float32, pin_memory=True)
                L.stage = d["stage"].data_ptr()
                d["stage_desc"] = torch.zeros((L.stage_rows * 2 * P.ceil_div(self.cap, P.KB) * 5,), dtype=torch.int32, pin_memory=True)
                L.stage_desc = d["stage_desc"].data_ptr() if self.spectral else None
            d["bank"], d["cap"] = bank.data, self.cap
            L.bank, L.bank_unit_stride, L.bank_chan_stride, L.cap = bank.data.data_ptr(), bank.data.stride(0), bank.data.stride(1), self.cap
            L.dev_len = bank.lengths.data_ptr()
        L.host_len, L.clipped = self.host_len.ctypes.data, self._clipped.ctypes.data
        L.used, L.use_seq = self._used.ctypes.data, self._use_seq.ctypes.data      # (eviction inside the call: _take_slots' policy)
        L.spec_stale = self._stale.ctypes.data if self.spectral else None
        L.keep = -1 if self.truncate_to is None else int(self.truncate_to)
        if d["dirs"] is not table_dirs or len(table_dirs) != L.n_table_dirs:
            d["dirs"] = table_dirs
            d["dirs_arr"] = (ctypes.c_char_p * max(1, len(table_dirs)))(*[os.fsencode(p_) for p_ in table_dirs])
            L.table_dirs, L.n_table_dirs = d["dirs_arr"], len(table_dirs)
        if d["pk"] is not pair_keys:
            d["pk"], d["ps"] = pair_keys, pair_slots
            L.pair_keys, L.pair_slots, L.pair_cap = pair_keys.ctypes.data, pair_slots.ctypes.data, int(pair_keys.shape[0])
        if d["free_ver"] != self._free_ver or L.n_free != d["free_n"] or (L.n_free < 16 and len(self._free) > L.n_free):
            top = self._free[-self._LOADER_FREE:]
            d["free"][:len(top)] = top                           # (stack order: the library pops from the end, as _take_slots does)
            L.n_free = d["free_n"] = len(top)
            d["free_ver"] = self._free_ver

    def adopt_loaded(self, d, key_of=None, keys=None) -> int:
        """Book what ``ss_ctx_observe_requests_load`` / ``ss_ctx_load_rir_files`` loaded (loader dict `d`): entries leave the free
        list and are bound to `key_of(pair_key)` (or to `keys[i]`: the file form) exactly as ``load_files`` binds them.  Returns
        the number of rows adopted."""
        L = d["s"]
        k = int(L.n_loaded)
        if k == 0:
            return 0
        slots = d["loaded_slot"][:k].tolist()
        ke = int(L.n_evicted)
        if ke:                                                   # entries the library reused: their keys leave the store's books
            tell = self.on_evict is not None or bool(self._evict_hooks)
            for slot in d["evicted"][:ke].tolist():
                victim = self._key_at[slot]
                del self._slot_of[victim]
                self._key_at[slot] = None
                self._used[slot] = False
                self._pending.pop(slot, None)
                if tell:
                    self._notify_evict(victim, slot)
            L.n_evicted = 0
        kf = k - ke                                              # ... the others came off the free stack
        if kf:
            if self._free[-kf:][::-1] == slots[:kf]:             # the library pops the free stack from its end, as _take_slots does
                del self._free[-kf:]
            else:                                                # (the list changed under the loader's mirror: take them out one by one)
                gone = set(slots[:kf])
                self._free = [f for f in self._free if f not in gone]
                self._free_ver += 1
        d["free_n"] = int(L.n_free)
        for key, sl in zip(keys if keys is not None else map(key_of, d["loaded_key"][:k].tolist()), slots):
            self._bind(key, sl)
        sl_np = np.asarray(slots)
        self._dev_len[sl_np] = self.host_len[sl_np]              # (host_len / _clipped / _stale were written by the library)
        self.misses += k
        L.n_loaded = 0
        return k

    def touch_slots(self, slots: np.ndarray) -> None:
        """Column paths (``DeferredResolver``, tables of ``RirIndex``) look slots up without going through ``slot()``: this
        marks them as handed out for the current batch, so that a miss of the same step cannot evict them."""
        self._batch_of[slots] = self._batch
        if self.group > 1:                                       # recency and the in-use guard live on a group's FIRST slot
            self._batch_of[(np.asarray(slots) // self.group) * self.group] = self._batch

    def slot(self, key, loader, refresh: bool = False) -> int:
        """Bank slot of ``key`` (first slot of its group); ``loader()`` -> float array [L,2] / [2,L] or None (a list of
        ``group`` of them when group > 1) is only called on a miss (or always with ``refresh=True``: live RIRs that
        change every step keep their slot)."""
        if key in self._slot_of:
            slot = self._slot_of.pop(key)
            self._slot_of[key] = slot                   # most recently used
            self._seq += 1
            self._use_seq[slot] = self._seq
            # a row clipped while only 1-s clips existed is reloaded once whole RIRs are needed (truncate_to = None)
            if refresh or (self.truncate_to is None and self._clipped[slot:slot + self.group].any()):
                self._load_into(slot, loader())
            else:
                self.hits += 1
            self._batch_of[slot] = self._batch
            return slot
        self.misses += 1
        slot = self._take_slot()
        self._bind(key, slot)
        self._load_into(slot, loader())
        return slot

    def _load_into(self, slot: int, loaded) -> None:
        if self.group == 1:
            self._upload(slot, loaded)
            return
        loaded = list(loaded) if loaded is not None else [None] * self.group
        assert len(loaded) == self.group
        for k, r in enumerate(loaded):
            self._upload(slot + k, r)

    def slot_many(self, keys: Sequence, loaders: Sequence, workers: int = 8) -> List[int]:
        """Slots of many keys at once (scene load, `AudioGoalBatcher`): the misses' loaders run on a thread pool
        (file reads overlap), their rows are packed into ONE pinned staging block and reach the bank with one H2D copy
        plus one row scatter, instead of one 128 KB copy per file.  Same LRU semantics as `slot()`; a batch must fit the
        store (its slots are returned together, so none of them may evict another)."""
        from concurrent.futures import ThreadPoolExecutor
        G = self.group
        out: List[int] = [-1] * len(keys)
        todo = []
        if len(set(keys)) > self.slots // G:
            raise ValueError(f"slot_many: {len(set(keys))} distinct keys do not fit a store of {self.slots // G} entries")
        self.begin_batch()
        for i, key in enumerate(keys):
            if key in self._slot_of:
                out[i] = self.slot(key, loaders[i])
            else:
                todo.append(i)
        first = {}
        for i in todo:                                   # duplicates inside the batch load once
            first.setdefault(keys[i], i)
        uniq = list(first.values())
        chunk = max(1, 256 // G)
        for lo in range(0, len(uniq), chunk):
            part = uniq[lo:lo + chunk]
            if workers > 1 and len(part) > 1:
                with ThreadPoolExecutor(max_workers=workers) as pool:
                    loaded = list(pool.map(lambda i: loaders[i](), part))
            else:
                loaded = [loaders[i]() for i in part]
            rows = []
            for item in loaded:
                item = [item] if G == 1 else (list(item) if item is not None else [None] * G)
                assert len(item) == G
                rows += [_planar(r) for r in item]
            kept = [self._kept_len(r.shape[1]) for r in rows]
            self._ensure_cap(max(kept + [0]))
            stage = torch.zeros((len(rows), 2, self.cap), dtype=torch.float32, pin_memory=self.device.type == "cuda")
            stage_np = stage.numpy()
            lens = np.asarray(kept, np.int32)
            slots = []
            taken = self._take_slots(len(part))
            for j, i in enumerate(part):
                self.misses += 1
                sl = taken[j]
                self._bind(keys[i], sl)
                for g in range(G):
                    r, n = rows[j * G + g], kept[j * G + g]
                    stage_np[j * G + g, :, :n] = r[:, :n]
                    slots.append(sl + g)
            idx = torch.as_tensor(slots, dtype=torch.long, device=self.device)
            self._about_to_write()
            self.bank.data.index_copy_(0, idx, stage.to(self.device, non_blocking=True))
            self.bank.lengths.index_copy_(0, idx, torch.from_numpy(lens).to(self.device))
            self._dev_len[np.asarray(slots)] = lens
            self.host_len[np.asarray(slots)] = lens
            self._stale[np.asarray(slots)] = True
            self._clipped[np.asarray(slots)] = [n < r.shape[1] for n, r in zip(kept, rows)]
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).synchronize()   # the pinned block dies with this scope
        for i in todo:
            out[i] = self._slot_of[keys[i]]
        return out

    # ---- float32 wav files, read natively (ss_wav_read_rirs_f32) ----------------------------------------------------------
    _FILE_CHUNK = 256                                            # rows per staging block (two blocks: read k+1 under copy k)

    scatter_from_host = True      # the scatter kernel reads the pinned staging block itself (False: one H2D copy first)

    def _scatter_staged(self, stage, pidx, plen, n_rows: int):
        """Rows [0, n_rows) of the staging block `stage` ([*, cap, 2] wav layout) -> bank rows pidx[i] with lengths plen[i]
        (int32 tensors next to the block: pinned on a GPU store).  GPU stores: ONE launch of the library's scatter
        (ss_bank_scatter_rows_f32 - it reads block, slots and lengths from the pinned memory itself, transposes into the
        planar rows and writes the length table); returns the event behind it (the block may be refilled after it).
        Host stores (tests without a GPU): the same scatter in torch."""
        self._about_to_write()
        if self.device.type != "cuda":
            idx = pidx[:n_rows].long()
            self.bank.data.index_copy_(0, idx, stage[:n_rows].permute(0, 2, 1))
            self.bank.lengths.index_copy_(0, idx, plen[:n_rows])
            return None
        from . import _lib
        data = self.bank.data
        di = self.device.index if self.device.index is not None else torch.cuda.current_device()
        src = stage if self.scatter_from_host else stage[:n_rows].to(self.device, non_blocking=True)

        def launch():
            _lib.check(_lib.load().ss_bank_scatter_rows_f32(
                src.data_ptr(), 2 * self.cap, pidx.data_ptr(), plen.data_ptr(), n_rows, data.data_ptr(), data.stride(0),
                data.stride(1), self.cap, self.bank.lengths.data_ptr(), torch._C._cuda_getCurrentRawStream(di)),
                "ss_bank_scatter_rows_f32")
            ev = torch.cuda.Event()
            ev.record()
            return ev
        if torch.cuda.current_device() == di:
            return launch()
        with torch.cuda.device(di):
            return launch()

    def _file_stage(self, k: int):
        """pinned [_FILE_CHUNK, cap, 2] staging block k & 1 (wav layout), free to be overwritten"""
        if getattr(self, "_fstage", None) is None or self._fstage[0].shape[1] != self.cap:
            #  the scatter of an earlier chunk may still be READING the old pinned blocks (scatter_from_host: the
            # kernel pulls them over the host link itself, and torch's caching host allocator knows nothing of that read): wait
            # for every outstanding scatter before the blocks are dropped, or the next torch.zeros() hands them out zero-filled
            for ev in getattr(self, "_fstage_ev", None) or ():
                if ev is not None:
                    ev.synchronize()
            pin = self.device.type == "cuda"
            self._fstage = [torch.zeros((self._FILE_CHUNK, self.cap, 2), dtype=torch.float32, pin_memory=pin) for _ in range(2)]
            # the rows' bank slots and lengths travel the same way (a list -> device tensor conversion is a SYNCHRONOUS pageable
            # copy: two of them were a third of a step that loads one new pose)
            self._fstage_idx = [torch.zeros((self._FILE_CHUNK,), dtype=torch.int32, pin_memory=pin) for _ in range(2)]
            self._fstage_len = [torch.zeros((self._FILE_CHUNK,), dtype=torch.int32, pin_memory=pin) for _ in range(2)]
            self._fstage_ev = [None, None]
        if self._fstage_ev[k & 1] is not None:
            self._fstage_ev[k & 1].synchronize()                 # the H2D copy that last read this block has run
            self._fstage_ev[k & 1] = None
        return self._fstage[k & 1]

    def load_files(self, keys: Sequence, paths: Sequence, reader=None, missing_ok: bool = False, threads: int = 0,
                   new_batch: bool = True) -> List[int]:
        """Slots of many keys whose RIRs are wav FILES: `paths[i]` is the file of `keys[i]` (group == 1) or the list of its
        `group` files (None = no such file: a zero row).  The misses are read by the library's own reader
        (ss_wav_read_rirs_f32: RIFF header parsed in C++, the first `truncate_to` frames read() straight into a pinned
        staging block in the file's own interleaved layout, plain threads - no scipy, no per-file arrays, no host
        transpose, no GIL), cross PCIe as one copy per block of 256 rows and are transposed into the planar bank rows by the
        scatter on the device.  Reference semantics (simulator.py:615-624) are kept file by file: whatever is not a plain
        float32 stereo wav goes through `reader(path)` (default ``sim_audio.wav_rir_reader``: scipy, ValueError -> zero
        RIR), an empty file is the zero RIR, a file that cannot be opened raises FileNotFoundError unless `missing_ok`.
        Same LRU semantics as ``slot_many``; nothing of the store changes when a file raises.  `new_batch=False`: the loads
        belong to the batch the caller has open (a step's pose misses: the rows that step already looked up stay protected)."""
        from . import _lib
        G = self.group
        if reader is None:
            from .sim_audio import wav_rir_reader as reader
        out: List[int] = [-1] * len(keys)
        first = {}
        if new_batch:
            self.begin_batch()
        for i, key in enumerate(keys):
            if key in self._slot_of:
                plist = [paths[i]] if G == 1 else list(paths[i])
                out[i] = self.slot(key, (lambda pl=plist: (reader(pl[0]) if G == 1 else [reader(p) if p else None for p in pl])))
            else:
                first.setdefault(key, i)
        uniq = list(first.values())
        if len(set(keys)) > self.slots // G:
            raise ValueError(f"load_files: {len(set(keys))} distinct keys do not fit a store of {self.slots // G} entries")
        keep = -1 if self.truncate_to is None else int(self.truncate_to)
        per = max(1, self._FILE_CHUNK // G)
        for c, lo in enumerate(range(0, len(uniq), per)):
            part = uniq[lo:lo + per]
            flat = []                                             # (row in the block, path) of the files that exist
            for j, i in enumerate(part):
                for g, pth in enumerate([paths[i]] if G == 1 else list(paths[i])):
                    if pth:
                        flat.append((j * G + g, pth))
            n_rows = len(part) * G
            while True:                                           # (again after the rows have grown)
                stage = self._file_stage(c)
                snp = stage.numpy()
                dense = len(flat) == n_rows
                tgt = snp if dense else np.zeros((max(len(flat), 1), self.cap, 2), np.float32)
                kept, frames, status = _lib.wav_read_rirs([p for _, p in flat], tgt, self.cap, keep=keep, threads=threads)
                too_long = status == _lib.WAV_TOO_LONG
                if too_long.any():
                    self._ensure_cap(int(self._kept_len(int(frames[too_long].max()))))
                    continue
                # scipy's semantics for everything unusual (int16 / mono / extensible headers ...): read by `reader` here, so that a
                # file longer than the rows grows them like a float32 file does (found by scripts/gpu_fuzz_plugin.py: an int16
                # RIR of 1.5 s in whole-RIR mode raised instead)
                by_reader = {f: _planar(reader(flat[f][1])) for f in np.flatnonzero(status == _lib.WAV_UNSUPPORTED)}
                need = max([self._kept_len(r.shape[1]) for r in by_reader.values()] + [0])
                if need > self.cap:
                    self._ensure_cap(int(need))
                    continue
                break
            lens = np.zeros((n_rows,), np.int32)
            full = np.zeros((n_rows,), np.int32)
            if not dense:
                snp[:n_rows] = 0.0
            for f, (row, pth) in enumerate(flat):
                st = int(status[f])
                if st == _lib.WAV_MISSING and not missing_ok:
                    raise FileNotFoundError(pth)
                if st == _lib.WAV_UNSUPPORTED:
                    r = by_reader[f]
                    n = self._kept_len(r.shape[1])
                    snp[row, :n, :] = r[:, :n].T
                    snp[row, n:, :] = 0.0
                    lens[row], full[row] = n, r.shape[1]
                    continue
                if not dense:
                    snp[row] = tgt[f]
                lens[row], full[row] = kept[f], frames[f]
            taken = self._take_slots(len(part))
            self.misses += len(part)
            for i, sl in zip(part, taken):
                self._bind(keys[i], sl)
            sl_np = np.asarray(taken) if G == 1 else (np.asarray(taken)[:, None] + np.arange(G)[None, :]).reshape(-1)
            pidx, plen = self._fstage_idx[c & 1], self._fstage_len[c & 1]
            pidx.numpy()[:n_rows] = sl_np
            plen.numpy()[:n_rows] = lens
            self._fstage_ev[c & 1] = self._scatter_staged(stage, pidx, plen, n_rows)
            self._dev_len[sl_np] = lens
            self.host_len[sl_np] = lens
            self._stale[sl_np] = True
            self._clipped[sl_np] = lens < full
        for i, key in enumerate(keys):
            if out[i] < 0:
                out[i] = self._slot_of[key]
        return out


class BucketedRirStore:
    """``RirStore`` over a LENGTH-BUCKETED bank: one sub-store per bucket, each with its own fixed row
    capacity and its own LRU; a key lives in the smallest bucket that holds its (longest) RIR.  Global slot = the
    bucket's first index + the sub-store's slot, so unit descriptors and ``host_len`` work as with one bank.

    What it fixes against the single-capacity store: a long RIR (an SS2.0 ray-traced 3-s response, one long wav among
    short ones) no longer reallocates and copies the whole bank at the new capacity in the middle of an episode, does not
    multiply the HBM of every slot, and launches whose units all sit in bucket 0 keep the loop-free kernel.  Only the
    LAST bucket may still grow (up to ``max_cap``), and then only its own rows move.

    ``caps`` ascending (samples; bucket 0 <= one partition block keeps the loop-free kernel), ``slots[b]`` entries each."""

    def __init__(self, slots: Sequence[int], caps: Sequence[int], device, truncate_to: Optional[int] = None,
                 max_cap: int = 1 << 18, on_grow=None, group: int = 1, spectral: bool = False):
        assert len(slots) == len(caps) and 1 <= len(caps) <= 4 and list(caps) == sorted(caps)
        self.device = torch.device(device)
        self.group, self.on_grow, self.spectral = group, on_grow, spectral
        self.first = [int(v) for v in np.cumsum([0] + list(slots[:-1]))]
        total = int(sum(slots))
        lengths = torch.zeros((total,), dtype=torch.int32, device=self.device)
        self.host_len = np.zeros((total,), np.int32)
        self.stores: List[RirStore] = []
        for b, (n, cap) in enumerate(zip(slots, caps)):
            last = b == len(caps) - 1
            st = RirStore(n, cap, device, truncate_to=truncate_to, max_cap=max_cap if last else cap + (cap & 1),
                          on_grow=self._sub_grown, group=group, spectral=spectral)
            st.bank.lengths = lengths[self.first[b]:self.first[b] + n]        # views of the one global array
            st.host_len = self.host_len[self.first[b]:self.first[b] + n]
            self.stores.append(st)
        self.bank = BucketedRirBank([st.bank for st in self.stores], lengths, self.first)
        self.slots = total
        self._where: Dict[object, int] = {}

    # -- the RirStore interface the engine / loaders use
    @property
    def truncate_to(self):
        return self.stores[0].truncate_to

    @truncate_to.setter
    def truncate_to(self, v):
        for st in self.stores:
            st.truncate_to = v

    @property
    def cap(self) -> int:
        return max(st.cap for st in self.stores)

    @property
    def hits(self) -> int:
        return sum(st.hits for st in self.stores)

    @property
    def misses(self) -> int:
        return sum(st.misses for st in self.stores)

    @property
    def grown(self) -> int:
        return sum(st.grown for st in self.stores)

    def _sub_grown(self, _bank) -> None:                          # (only the last bucket can: its rows alone moved)
        self.bank = BucketedRirBank([st.bank for st in self.stores], self.bank.lengths, self.first)
        if self.on_grow is not None:
            self.on_grow(self.bank)

    def begin_batch(self) -> None:
        for st in self.stores:
            st.begin_batch()

    def clear(self) -> None:
        for st in self.stores:
            st.clear()
        self._where.clear()

    def sync_spectra(self) -> int:
        return sum(st.sync_spectra() for st in self.stores)

    def flush_uploads(self) -> int:
        return sum(st.flush_uploads() for st in self.stores)

    def _bucket_for(self, loaded) -> int:
        items = [loaded] if self.group == 1 else (list(loaded) if loaded is not None else [None])
        n = max([self.stores[0]._kept_len(_planar(r).shape[1]) for r in items] + [0])
        for b, st in enumerate(self.stores[:-1]):
            if n <= st.cap:
                return b
        return len(self.stores) - 1

    def slot(self, key, loader, refresh: bool = False) -> int:
        b = self._where.get(key)
        if b is not None and key not in self.stores[b]._slot_of:  # evicted from its bucket meanwhile
            b = None
        if b is not None and not refresh:
            st = self.stores[b]
            clipped_reload = st.truncate_to is None and st._clipped[st._slot_of[key]:st._slot_of[key] + self.group].any()
            if not clipped_reload:
                return self.first[b] + st.slot(key, loader, False)
        loaded = loader()                                          # miss, live refresh, or a row that must be re-read whole
        nb = self._bucket_for(loaded)
        if b is not None and b != nb:                              # the key changes length class: leave the old bucket
            old = self.stores[b]
            old.release(key)
        self._where[key] = nb
        st = self.stores[nb]
        return self.first[nb] + st.slot(key, lambda: loaded, refresh=key in st._slot_of)

    def slot_many(self, keys: Sequence, loaders: Sequence, workers: int = 8) -> List[int]:
        """Bulk load: files are read in parallel, then routed bucket by bucket through the sub-stores' own ``slot_many``
        (one H2D block copy per bucket)."""
        from concurrent.futures import ThreadPoolExecutor
        out: List[int] = [-1] * len(keys)
        todo = []
        for i, key in enumerate(keys):
            b = self._where.get(key)
            if b is not None and key in self.stores[b]._slot_of:
                st = self.stores[b]
                sl = st._slot_of[key]
                if st.truncate_to is None and st._clipped[sl:sl + self.group].any():
                    # a row clipped while only 1-s clips existed must be re-read WHOLE, and the whole RIR may belong to a
                    # longer bucket: through slot(), which re-buckets it (the sub-store's own reload raised
                    # 'exceeds max_cap' for a multi-second RIR sitting in bucket 0)
                    out[i] = self.slot(key, loaders[i])
                else:
                    out[i] = self.first[b] + st.slot(key, loaders[i])
            else:
                todo.append(i)
        uniq = list({keys[i]: i for i in reversed(todo)}.values())[::-1]
        if workers > 1 and len(uniq) > 1:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                loaded = dict(zip(uniq, pool.map(lambda i: loaders[i](), uniq)))
        else:
            loaded = {i: loaders[i]() for i in uniq}
        per_bucket: Dict[int, List[int]] = {}
        for i in uniq:
            per_bucket.setdefault(self._bucket_for(loaded[i]), []).append(i)
        for b, idx in per_bucket.items():
            got = self.stores[b].slot_many([keys[i] for i in idx], [(lambda i=i: loaded[i]) for i in idx], workers=1)
            for i, sl in zip(idx, got):
                self._where[keys[i]] = b
                out[i] = self.first[b] + sl
        for i in todo:
            if out[i] < 0:
                b = self._where[keys[i]]
                out[i] = self.first[b] + self.stores[b]._slot_of[keys[i]]
        return out


    def load_files(self, keys: Sequence, paths: Sequence, reader=None, missing_ok: bool = False, threads: int = 0,
                   new_batch: bool = True) -> List[int]:
        """``RirStore.load_files`` over the length buckets: the files' frame counts are probed first (the library's reader with
        a one-frame row: header only, nothing copied), every key goes to the smallest bucket that holds its (longest) RIR and
        each bucket loads its files through its sub-store's native path."""
        from . import _lib
        G = self.group
        out: List[int] = [-1] * len(keys)
        todo = []
        for i, key in enumerate(keys):
            b = self._where.get(key)
            if b is not None and key in self.stores[b]._slot_of:
                st = self.stores[b]
                sl = st._slot_of[key]
                if st.truncate_to is None and st._clipped[sl:sl + G].any():
                    plist = [paths[i]] if G == 1 else list(paths[i])
                    rd = reader
                    if rd is None:
                        from .sim_audio import wav_rir_reader as rd
                    out[i] = self.slot(key, (lambda pl=plist, rd=rd: rd(pl[0]) if G == 1 else [rd(p) if p else None for p in pl]))
                else:
                    out[i] = self.first[b] + st.slot(key, lambda: None)
            else:
                todo.append(i)
        first = {}
        for i in todo:
            first.setdefault(keys[i], i)
        uniq = list(first.values())
        flat = [(i, p) for i in uniq for p in ([paths[i]] if G == 1 else list(paths[i])) if p]
        probe = np.zeros((max(len(flat), 1), 1, 2), np.float32)
        _kept, frames, status = _lib.wav_read_rirs([p for _, p in flat], probe, 1, keep=-1, threads=threads)
        longest: Dict[int, int] = {i: 0 for i in uniq}
        keep0 = self.stores[0]._kept_len
        for (i, pth), fr, st_ in zip(flat, frames, status):
            n = int(fr)
            if st_ == _lib.WAV_UNSUPPORTED:                       # odd file: the Python reader decides (and gives its length)
                rd = reader
                if rd is None:
                    from .sim_audio import wav_rir_reader as rd
                n = _planar(rd(pth)).shape[1]
            longest[i] = max(longest[i], keep0(n))
        per_bucket: Dict[int, List[int]] = {}
        for i in uniq:
            b = len(self.stores) - 1
            for bb, st in enumerate(self.stores[:-1]):
                if longest[i] <= st.cap:
                    b = bb
                    break
            per_bucket.setdefault(b, []).append(i)
        for b, idx in per_bucket.items():
            got = self.stores[b].load_files([keys[i] for i in idx], [paths[i] for i in idx], reader=reader, missing_ok=missing_ok,
                                            threads=threads, new_batch=new_batch)
            for i, sl in zip(idx, got):
                self._where[keys[i]] = b
                out[i] = self.first[b] + sl
        for i in todo:
            if out[i] < 0:
                b = self._where[keys[i]]
                out[i] = self.first[b] + self.stores[b]._slot_of[keys[i]]
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
    native = _native_wav(reader) and hasattr(store, "load_files")
    for lo in range(0, len(paths), batch):                      # threaded reads, one H2D copy per batch
        part = paths[lo:lo + batch]
        if native:                                              # the library's own reader (ss_wav_read_rirs_f32): no scipy
            store.load_files(part, part, reader=reader, threads=workers)
        else:
            store.slot_many(part, [(lambda path=path: reader(path)) for path in part], workers=workers)
    return len(paths)


def _native_wav(reader) -> bool:
    """`reader` is the stock wav reader (sim_audio.wav_rir_reader, possibly through functools.partial(lenient=...)): its
    files can go through the library's native reader, which falls back to it for anything unusual"""
    import functools
    from .sim_audio import wav_rir_reader
    return reader is wav_rir_reader or (isinstance(reader, functools.partial) and reader.func is wav_rir_reader)


class AudioEngine:
    """Renderer + RIR store + source registry: what ``ss_amd.sim_audio`` talks to (one per process / GPU).

    RIR rows: as long as every registered clip is a 1-s clip (SoundSpaces 1.0 default sounds) only h[0:sr] can reach
    the observation (simulator.py:629-632), so rows are clipped to sr samples and the loop-free kernel runs.  The first
    multi-second clip (simulator.py:634-647: the window also hears the reverb tail of earlier seconds) or an SS2.0
    renderer (``step_time``) switches the store to full-length rows: rows that had been clipped reload at their next
    use, longer RIRs grow the bank (RirStore)."""

    def __init__(self, sampling_rate: int, device="cuda", rir_slots: int = 4096, rir_cap: Optional[int] = None,
                 rir_max_cap: int = 1 << 18, rir_group: int = 1, rir_spectral: Optional[bool] = None,
                 rir_buckets: Optional[Sequence[Tuple[int, int]]] = None, spectral_hbm_fraction: float = 0.5,
                 spectral_max_units: int = 0, **renderer_kwargs):
        """rir_spectral: keep the RIR rows' block spectra in HBM as well (2x the bytes per row) and run k_conv_spec /
        k_obs_rows<SPECTRAL> (no forward FFT per step): for STATIC banks (SoundSpaces 1.0 RIR files); live SS2.0 RIRs change
        every step and stay on the time-domain kernels.  None (default) = decided here: ON for file-backed stores at rates
        whose rows span several partition blocks (44.1 / 48 kHz - the reference's Replica rate,
        configs/audionav/av_nav/replica/audiogoal.yaml:18: every observation is three forward FFTs per ear and a stash round
        trip there; cfg[2]: 271 vs 334 us per 512 units) when rows + spectra fit `spectral_hbm_fraction` of the device's free
        memory.  Round 6: at 16 kHz too, under the same condition, for EVERY launch: the fused kernel is faster from the spectral
        rows at every size (same box, alternating, profiles/r6/kbench_bank_form_16k.txt: 1 / 32 / 64 / 128 / 256 / 512 / 2048 units
        17.1 / 19.0 / 20.3 / 25.0 / 48.5 / 96.2 / 365 us from the time-domain rows, 12.9 / 15.6 / 17.4 / 23.2 / 44.8 / 86.3 / 325 us
        from the spectral rows) - what it costs is HBM (rows + spectra = 3 x the rows) and one transform per loaded row, which is why
        it is tied to `spectral_hbm_fraction`.  `spectral_max_units` > 0 restores a per-launch choice: launches of more units than
        that (without distractor terms) read the time-domain rows, and the block spectra of freshly loaded rows are only built
        when a launch needs them."""
        self.renderer = BatchedAudioRenderer(sampling_rate, device=device, **renderer_kwargs)
        self._native_readers: Dict[int, tuple] = {}              # rir_file_slot: id(reader) -> (stock wav reader?, lenient?, reader)
        self._file_loader = None                                 # RirStore.miss_loader dict of rir_file_slot (ss_ctx_load_rir_files)
        full = self.renderer.n_valid != self.renderer.sr or self.renderer.wrap
        self.spectral_max_units = 0
        if rir_spectral is None:
            rir_spectral = self._auto_spectral(sampling_rate, rir_slots if not rir_buckets else sum(b[0] for b in rir_buckets),
                                               rir_cap or sampling_rate, full, spectral_hbm_fraction)
            if rir_spectral and sampling_rate <= P.KB:
                self.spectral_max_units = int(spectral_max_units)
        self.renderer.spectral_max_units = self.spectral_max_units
        self.rir_spectral = bool(rir_spectral) and not full
        if rir_buckets:
            # length-bucketed bank: [(slots, cap samples), ...] ascending, e.g. [(4096, 16000), (256, 49152), (64, 65536)]
            self.store = BucketedRirStore([b[0] for b in rir_buckets], [b[1] for b in rir_buckets], self.renderer.device,
                                          truncate_to=None if full else int(sampling_rate), max_cap=rir_max_cap,
                                          on_grow=self.renderer.set_rir_bank, group=rir_group,
                                          spectral=rir_spectral and not full)
            self.renderer.set_rir_bank(self.store.bank)
            return
        self.store = RirStore(rir_slots, rir_cap or sampling_rate, self.renderer.device,
                              truncate_to=None if full else int(sampling_rate), max_cap=rir_max_cap,
                              on_grow=self.renderer.set_rir_bank, group=rir_group, spectral=rir_spectral and not full)
        self.store.defer_uploads = True            # single-row uploads of a step travel as one block (flushed before every launch)
        self.renderer.set_rir_bank(self.store.bank)

    def _auto_spectral(self, sr: int, slots: int, cap: int, full: bool, fraction: float) -> bool:
        dev = self.renderer.device
        if full or dev.type != "cuda":
            return False
        blocks = P.ceil_div(cap + (cap & 1), P.KB)
        need = slots * 2 * (cap * 4 + blocks * P.SPEC_FLOATS * 4)          # time-domain rows + their block spectra
        free, _total = torch.cuda.mem_get_info(dev)
        return need <= fraction * free

    def source_id(self, name: str, clip: np.ndarray) -> int:
        if self.store.truncate_to is not None and np.shape(clip)[0] != self.renderer.sr:
            self.store.truncate_to = None          # from now on whole RIRs; clipped rows reload at their next use
        sid = self.renderer.add_source(name, clip)
        ctx = getattr(self, "_ctx", None)
        if ctx is not None and name not in ctx._names:
            cid = ctx.add_source(name, self.renderer.sources._host[sid])
            assert cid == sid, "the context's sound ids mirror the renderer's"
        return sid

    # ---- the C++ context over the same banks (column paths: DeferredResolver) ---------------------------------------
    def context(self):
        """An ``ss_amd.context.AudioContext`` (planner, window cache and descriptor ring inside libss_hip.so) over THIS
        engine's source registry and RIR store: same sound ids, same bank slots.  Callers hand it unit COLUMNS
        (``observe_columns``); the per-unit Python planner of ``observe()`` is not involved."""
        if getattr(self, "_ctx", None) is None:
            from .context import AudioContext
            r = self.renderer
            ctx = AudioContext(r.sr, n_valid=r.n_valid, wrap=r.wrap, pad_mode=r.pad_mode)
            if self.spectral_max_units:
                ctx.set_spectral_policy(self.spectral_max_units)
            for sid, (name, _) in enumerate(sorted(r.sources.names.items(), key=lambda kv: kv[1])):
                assert ctx.add_source(name, r.sources._host[sid]) == sid
            self._ctx = ctx
            self._ctx_bank = None
            grow = self.store.on_grow

            def on_grow(bank):
                if grow is not None:
                    grow(bank)
                self._ctx_bank = None
            self.store.on_grow = on_grow
            # overlap mode (ctx.set_overlap): whatever the store writes to the bank goes behind the steps in flight on the lanes
            # (AudioContext.join is a no-op on a single-stream context: one attribute test per write)
            for st in getattr(self.store, "stores", [self.store]):
                st.before_device_write = ctx.join
        return self._ctx

    def _sync_context_bank(self, n_units: int = 0, distractor: bool = True):
        """n_units: size of the step about to be launched (0 = unknown).  Under the small-step policy (spectral_max_units) the
        block spectra of freshly loaded rows are only built when a launch is going to read them: large steps (which take the
        time-domain rows) skip the transform, the rows stay marked and are transformed before the next small step."""
        ctx = self.context()
        if n_units <= 0 or self.renderer._spectral_for(n_units, distractor):
            n_sync = self.store.sync_spectra()
        else:
            n_sync = 0
            self.store.flush_uploads()                             # (sync_spectra's first half: queued row uploads go out)
        bank = self.store.bank
        cur = self._ctx_bank                                     # (the tensors the context was last pointed at: identity, not
        if isinstance(bank, BucketedRirBank):                    # length buckets: ss_ctx_set_rir_buckets (every bucket's rows, and
            now = tuple(b.data for b in bank.banks) + tuple(b.spectra for b in bank.banks)      # its block spectra when kept)
            if n_sync or cur is None or len(cur) != len(now) or any(a is not b for a, b in zip(cur, now)):
                ctx.set_rir_buckets(bank, spectral=bool(self.store.spectral and bank.spectra))
                self._ctx_bank = now
            return ctx
        if n_sync or cur is None or cur[0] is not bank.data or cur[1] is not bank.spectra:   # two data_ptr() calls per step)
            ctx.set_rir_bank(bank.data, bank.lengths)
            if bank.spectra is not None and self.store.spectral:
                ctx.set_rir_spectra(bank.spectra)
            self._ctx_bank = (bank.data, bank.spectra)
        return ctx

    def observe_requests(self, recs: bytes, n: int, tables, spectrogram_out=None, audiogoal_out=None, loader=None) -> int:
        """One step from the packed request records of ``ss_amd.deferred`` (``ss_ctx_observe_requests``: lookups + planner +
        launch in one C call).  Returns the number of unresolved requests (0: the step is on the stream).  `loader` (a
        ``RirStore.miss_loader`` dict): poses that are not resident are loaded inside the same call when the library's fast
        path covers them (``ss_ctx_observe_requests_load``); the caller then books them (``RirStore.adopt_loaded``)."""
        ctx = self._sync_context_bank(n, getattr(self, "_req_has_distractor", True))
        if getattr(self, "_req_miss", None) is None or self._req_miss["buf"].shape[0] < n:
            import ctypes
            buf, cnt = np.zeros((max(n, 256),), np.int32), ctypes.c_int(0)
            self._req_miss = dict(buf=buf, n=cnt, ptr=buf.ctypes.data, n_ptr=ctypes.addressof(cnt))
        dev = self.renderer.device
        stream = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
        sg = None if spectrogram_out is None else spectrogram_out.data_ptr()
        ag = None if audiogoal_out is None else audiogoal_out.data_ptr()
        if loader is not None:
            call = lambda: ctx.observe_requests_load(recs, n, tables, loader, sg, ag, stream, self._req_miss)   # noqa: E731
        else:
            call = lambda: ctx.observe_requests(recs, n, tables, sg, ag, stream, self._req_miss)                # noqa: E731
        if torch.cuda.current_device() == (dev.index or 0):
            return call()
        with torch.cuda.device(dev):
            return call()

    def observe_columns(self, cols: Dict[str, np.ndarray], spectrogram_out=None, audiogoal_out=None) -> None:
        """One step from unit columns {sound, t0, rir[, dis_sound, dis_rir, last_rir, wrap, last_wrap]} (numpy, one entry
        per env; rir < 0 = silent) through the context: ONE ctypes call, outputs written into the given device tensors."""
        ctx = self._sync_context_bank(len(cols["sound"]), "dis_rir" in cols)
        ctx.observe(spectrogram_out=spectrogram_out, audiogoal_out=audiogoal_out, **cols)

    def begin_batch(self) -> None:
        self.store.begin_batch()

    def rir_slot(self, key, loader, refresh: bool = False) -> int:
        return self.store.slot(key, loader, refresh)

    def rir_file_slot(self, path: str, reader) -> int:
        """Bank entry of the RIR FILE `path` (keyed by the path, as ``rir_slot(path, ...)``).  A file that is not resident is read by
        the library's own reader when `reader` is the stock wav reader (``RirStore.load_files``: header parsed in C++, the frames
        read() straight into a pinned block, one scatter launch; anything unusual goes through `reader`, i.e. scipy with the
        reference's ValueError -> zero-RIR fallback, simulator.py:617-624) instead of scipy + a host transpose + a row upload: what
        an eager call pays on EVERY step of an agent that moves (simulator.py:615-618 reads the file on every cache-missing step)."""
        store = self.store
        if path in store._slot_of or getattr(store, "group", 1) != 1 or not hasattr(store, "load_files"):
            return store.slot(path, lambda: reader(path))
        native = self._native_readers.get(id(reader))
        if native is None:
            import functools
            native = self._native_readers[id(reader)] = (_native_wav(reader), isinstance(reader, functools.partial) and
                                                         bool(reader.keywords.get("lenient")), reader)
        if not native[0]:
            return store.slot(path, lambda: reader(path))
        if type(store) is RirStore and store.device.type == "cuda" and not native[1] and hasattr(store, "_batch_of"):
            # ONE C call (ss_ctx_load_rir_files): entry off the free stack (or the least recently used one), file read into the
            # store's pinned block, scatter launch, block spectra - the store's dictionaries follow from the report
            import ctypes
            ctx = self._sync_context_bank(1)
            d = self._file_loader
            if d is None:
                z = np.zeros((1,), np.int64)
                d = self._file_loader = store.miss_loader([], z, z.copy(), 0)
            else:
                store.refresh_loader(d, d["dirs"], d["pk"], d["ps"])
            dev = self.renderer.device
            stream = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
            rc = ctx.lib.ss_ctx_load_rir_files(ctx._h, d["ref"], (ctypes.c_char_p * 1)(os.fsencode(path)), 1,
                                               store._batch_of.ctypes.data, int(store._batch), int(store.slots), stream)
            if rc == 0:
                store.adopt_loaded(d, keys=[path])
                return store._slot_of[path]
            if rc < 0:
                ops._lib.check(rc, "ss_ctx_load_rir_files")
        return store.load_files([path], [path], reader=reader, missing_ok=native[1], new_batch=False)[0]

    def rir_len(self, slot: int) -> int:
        return int(self.store.host_len[slot])

    def observe(self, units: Sequence[UnitRequest], want_audiogoal: bool = False, want_spectrogram: bool = True,
                spectrogram_out=None, audiogoal_out=None) -> Dict[str, torch.Tensor]:
        if self.renderer._spectral_for(len(units), any(u.dis_rir >= 0 for u in units)):
            self.store.sync_spectra()
        else:
            self.store.flush_uploads()
        plan = self.renderer.plan(units)
        if not want_spectrogram:
            return {"audiogoal": self.renderer.render_audiogoal(plan, out=audiogoal_out)}
        ag, sg = self.renderer.render(plan, want_audiogoal=want_audiogoal, audiogoal_out=audiogoal_out,
                                      spectrogram_out=spectrogram_out)
        out = {"spectrogram": sg}
        if ag is not None:
            out["audiogoal"] = ag
        return out

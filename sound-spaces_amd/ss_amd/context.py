"""``AudioContext`` — Python face of the library's context API (include/ss_hip.h: ``ss_ctx_*``).

The planner, the bounded cache of source-window spectra and the pinned descriptor ring live INSIDE libss_hip.so
(csrc/ss_context.hpp); a vector step is ONE ctypes call taking numpy columns {sound, t0, rir, ...} of all envs.  This is
what the batched observers use (``VectorAudioObserver``, ``bench.py --path plugin``): no per-unit Python, no torch
tensor construction, no pageable H2D copy on the per-step path.  ``BatchedAudioRenderer.plan()/render()`` (descriptor
planning in Python, ``planning.py``) remains for pre-planned batches and as the executable specification the C++
planner is tested against (tests/test_context.py).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np

from . import _lib
from .planning import spectrogram_shape

_PAD = {"reflect": 0, "constant": 1, 0: 0, 1: 1}


def _col(a, dtype):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


_NO_MISS = np.zeros((0,), np.int32)


class AudioContext:
    def __init__(self, sampling_rate: int, step_time: Optional[float] = None, wrap: bool = False,
                 pad_mode="reflect", max_window_sets: int = 256, n_valid: Optional[int] = None):
        """step_time None: SoundSpaces 1.0 (1-s observations); step_time = STEP_TIME with wrap=True: SoundSpaces 2.0
        (n_valid: the samples per step given directly instead of as a fraction of a second)."""
        self.lib = _lib.load()
        self.sr = int(sampling_rate)
        self.n_valid = int(n_valid) if n_valid is not None else (self.sr if step_time is None else int(self.sr * step_time))
        self.wrap = bool(wrap)
        self.spectrogram_shape = spectrogram_shape(self.sr)
        h = ctypes.c_void_p()
        _lib.check(self.lib.ss_ctx_create(ctypes.byref(h), self.sr, self.n_valid, _PAD[pad_mode], int(self.wrap),
                                          int(max_window_sets)), "ss_ctx_create")
        self._h = h
        self.handle = AudioContext._next_handle          # torch.ops.ss_hip.ctx_observe(handle, ...) finds the context by it
        AudioContext._next_handle += 1
        AudioContext._by_handle[self.handle] = self
        self._native = False
        try:                                             # the C++ op layer resolves a handle to the raw ss_ctx* itself
            from . import ops
            if ops.NATIVE_OPS:
                import torch
                torch.ops.ss_hip.ctx_register(self.handle, int(h.value))
                self._native = True
        except ImportError:                              # (torch-free use of the context: planner tests)
            pass
        self._names: Dict[str, int] = {}
        self.lengths = []
        self._bank = None                    # keeps the borrowed tensors alive

    _next_handle = 1
    _by_handle = __import__("weakref").WeakValueDictionary()

    @staticmethod
    def from_handle(handle: int) -> "AudioContext":
        return AudioContext._by_handle[int(handle)]

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_native", False):
                try:
                    import torch
                    torch.ops.ss_hip.ctx_unregister(self.handle)
                except Exception:                        # interpreter shutdown
                    pass
            self.lib.ss_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- banks ---------------------------------------------------------------------------------------------
    def add_source(self, name: str, clip) -> int:
        """Register a mono clip (numpy float32 on the host, or a CUDA tensor); idempotent per name."""
        if name in self._names:
            return self._names[name]
        if hasattr(clip, "is_cuda"):
            import torch
            t = clip.to(torch.float32).contiguous().reshape(-1)
            if not t.is_cuda:
                raise _lib.SsHipError("add_source: tensors must live on the GPU (pass host clips as numpy arrays)")
            with torch.cuda.device(t.device):
                sid = self.lib.ss_ctx_add_source(self._h, t.data_ptr(), int(t.numel()), 1)
            n = int(t.numel())
        else:
            a = np.ascontiguousarray(clip, dtype=np.float32).reshape(-1)
            sid = self.lib.ss_ctx_add_source(self._h, a.ctypes.data, int(a.shape[0]), 0)
            n = int(a.shape[0])
        if sid < 0:
            _lib.check(sid, "ss_ctx_add_source")
        self._names[name] = sid
        self.lengths.append(n)
        return sid

    def add_source_len(self, name: str, length: int) -> int:
        """Planner-only registration (no device memory): for plan() on machines without a GPU."""
        if name in self._names:
            return self._names[name]
        sid = self.lib.ss_ctx_add_source_len(self._h, int(length))
        if sid < 0:
            _lib.check(sid, "ss_ctx_add_source_len")
        self._names[name] = sid
        self.lengths.append(int(length))
        return sid

    def set_rir_bank(self, data, lengths, interleaved: bool = False) -> None:
        """data: CUDA float32 [R,2,cap] (planar) or [R,cap,2] (wav-interleaved); lengths: CUDA int32 [R]."""
        if interleaved:
            R, cap, _ = data.shape
            us, cs, es = 2 * cap, 1, 2
        else:
            R, _, cap = data.shape
            us, cs, es = 2 * cap, cap, 1
        _lib.check(self.lib.ss_ctx_set_rir_bank(self._h, data.data_ptr(), lengths.data_ptr(), us, cs, es, int(cap)),
                   "ss_ctx_set_rir_bank")
        self._bank = (data, lengths)
        self.rir_cap = int(cap)
        self._spectra = None

    def set_rir_buckets(self, bank, spectral: bool = False) -> None:
        """A length-bucketed bank (``ss_amd.renderer.BucketedRirBank``; include/ss_hip.h ``ss_ctx_set_rir_buckets``):
        steps whose units all sit in bucket 0 keep the loop-free kernel, long RIRs live in buckets of their own.
        ``spectral``: use the buckets' spectral forms (every bucket must have one)."""
        arr = bank.c_array(bool(spectral))
        _lib.check(self.lib.ss_ctx_set_rir_buckets(self._h, ctypes.cast(arr, ctypes.c_void_p), len(bank.banks),
                                                   bank.lengths.data_ptr()), "ss_ctx_set_rir_buckets")
        self._bank = (bank, arr)
        self.rir_cap = int(bank.cap)
        self._spectra = None

    def set_rir_spectra(self, hspec) -> None:
        """Spectral form of the bank set by set_rir_bank() (ops.rir_spectra / RirBank.build_spectra): steps without a
        cross-fade then run k_conv_spec.  None switches back to the time-domain kernels."""
        if hspec is None:
            _lib.check(self.lib.ss_ctx_set_rir_spectra(self._h, None, 0), "ss_ctx_set_rir_spectra")
        else:
            _lib.check(self.lib.ss_ctx_set_rir_spectra(self._h, hspec.data_ptr(), int(hspec.shape[2])),
                       "ss_ctx_set_rir_spectra")
        self._spectra = hspec

    def set_overlap(self, n_streams: int = 2) -> None:
        """Consecutive observe() calls alternate between ``n_streams`` (2 .. 4) internal streams (ss_ctx_set_overlap): the
        load phase of step k+1 overlaps the STFT phase of step k; 3 pays for steps that fill at most half the chip (<= 64 envs),
        4 only with more than the runtime's default four hardware queues (GPU_MAX_HW_QUEUES=8).  Results are visible to a stream
        after ``join()``; steps in flight must write disjoint output rows, and bank rows that steps in flight may read must not
        be rewritten before a ``join()`` on the stream that carries the rewrite (an ``AudioEngine``'s stores do that themselves
        before every device write once the engine has handed out its context, and so do the library's in-call loaders; callers
        that write bank rows of their own join first)."""
        _lib.check(self.lib.ss_ctx_set_overlap(self._h, int(n_streams)), "ss_ctx_set_overlap")
        self.overlap = int(n_streams)

    def set_spectral_policy(self, max_units: int) -> None:
        """With both bank forms set: steps of more than ``max_units`` units of one-block rows (16 kHz) take the time-domain
        rows, smaller ones the spectral rows (ss_ctx_set_spectral_policy); 0 = the spectral form whenever it is set."""
        _lib.check(self.lib.ss_ctx_set_spectral_policy(self._h, int(max_units)), "ss_ctx_set_spectral_policy")

    def set_chip_share(self, n_sources: int) -> None:
        """This context is one of ``n_sources`` launch sources kept busy at once (e.g. two env groups stepped alternately, each
        with its own context and stream): its small steps split their rows over 1 / n_sources of the chip (ss_ctx_set_chip_share)."""
        _lib.check(self.lib.ss_ctx_set_chip_share(self._h, int(n_sources)), "ss_ctx_set_chip_share")

    def join(self, stream: Optional[int] = None) -> None:
        """Make `stream` (default: the current torch stream) wait for every step issued so far (no-op without overlap)."""
        if getattr(self, "overlap", 1) <= 1:
            return
        if stream is None:
            import torch
            stream = torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
        _lib.check(self.lib.ss_ctx_join(self._h, stream), "ss_ctx_join")

    def set_rir_cap_for_planning(self, cap: int) -> None:
        """plan()-only use without a GPU: the bank capacity decides how many partition blocks a key needs."""
        _lib.check(self.lib.ss_ctx_set_rir_bank(self._h, None, None, 2 * cap, cap, 1, int(cap)), "ss_ctx_set_rir_bank")
        self.rir_cap = int(cap)

    # ---- one step -------------------------------------------------------------------------------------------
    def _units(self, sound, t0, rir, dis_sound, dis_rir, last_rir, wrap, last_wrap):
        cols = [_col(sound, np.int32), _col(t0, np.int32), _col(rir, np.int32), _col(dis_sound, np.int32),
                _col(dis_rir, np.int32), _col(last_rir, np.int32), _col(wrap, np.uint8), _col(last_wrap, np.uint8)]
        n = int(cols[0].shape[0])
        for c in cols:
            assert c is None or c.shape == (n,)
        u = _lib.SsUnits(*[None if c is None else c.ctypes.data for c in cols])
        return u, n, cols

    def observe(self, sound, t0, rir, spectrogram_out=None, audiogoal_out=None, dis_sound=None, dis_rir=None,
                last_rir=None, wrap=None, last_wrap=None, stream: Optional[int] = None) -> None:
        """Render one step into the given CUDA tensors ([n,65,T4,2] and / or [n,2,sr], float32, contiguous) on the
        current torch stream (or `stream`, a raw hipStream_t).  rir < 0 = silent unit."""
        import torch
        u, n, keep = self._units(sound, t0, rir, dis_sound, dis_rir, last_rir, wrap, last_wrap)
        sg = ag = None
        dev = None
        if spectrogram_out is not None:
            assert spectrogram_out.is_cuda and spectrogram_out.dtype == torch.float32 and spectrogram_out.is_contiguous()
            assert tuple(spectrogram_out.shape) == (n,) + self.spectrogram_shape
            sg, dev = spectrogram_out.data_ptr(), spectrogram_out.device
        if audiogoal_out is not None:
            assert audiogoal_out.is_cuda and audiogoal_out.dtype == torch.float32 and audiogoal_out.is_contiguous()
            assert tuple(audiogoal_out.shape) == (n, 2, self.sr)
            ag, dev = audiogoal_out.data_ptr(), audiogoal_out.device
        if dev is None:
            raise ValueError("observe: pass spectrogram_out and / or audiogoal_out")
        if stream is None:
            stream = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
        with torch.cuda.device(dev):
            _lib.check(self.lib.ss_ctx_observe(self._h, ctypes.byref(u), n, ag, sg, stream), "ss_ctx_observe")

    def prepare(self, sound, t0, rir, dis_sound=None, dis_rir=None, last_rir=None, wrap=None, last_wrap=None):
        """Unit columns converted ONCE into the ss_units struct ``observe_prepared`` takes (callers that replay known steps -
        bench.py, tests - or keep their columns in place and only rewrite the values: the arrays are borrowed)."""
        u, n, keep = self._units(sound, t0, rir, dis_sound, dis_rir, last_rir, wrap, last_wrap)
        return dict(u=u, ref=ctypes.byref(u), n=n, keep=keep)

    def observe_prepared(self, prep, spectrogram_ptr, audiogoal_ptr, stream: int) -> None:
        """``observe`` without the per-call conversions: raw device pointers (int or None) and a raw hipStream_t; the
        current device must be the context's.  ~3 us of Python per call instead of ~30."""
        rc = self.lib.ss_ctx_observe(self._h, prep["ref"], prep["n"], audiogoal_ptr, spectrogram_ptr, stream)
        if rc != 0:
            _lib.check(rc, "ss_ctx_observe")

    @staticmethod
    def features(logmel_out=None, mel_start=None, mel_w=None, mel_eps: float = 1e-6, gccphat_out=None, max_lag: int = 32,
                 gcc_eps: float = 1e-8):
        """The extension features of a step as the C struct ``ss_features`` (device tensors, borrowed): pass the result to
        ``observe_prepared(..., features=)``.  logmel_out [n, n_mels, T, 2] with the band-sparse bank (mel_start int32
        [n_mels], mel_w float32 [n_mels, max_len]: ``planning.mel_filterbank_sparse``); gccphat_out [n, 2*max_lag+1, T]."""
        f = _lib.SsFeatures()
        if logmel_out is not None:
            f.logmel, f.mel_start, f.mel_w = logmel_out.data_ptr(), mel_start.data_ptr(), mel_w.data_ptr()
            f.n_mels, f.max_len, f.mel_eps = int(mel_w.shape[0]), int(mel_w.shape[1]), float(mel_eps)
        if gccphat_out is not None:
            f.gccphat, f.max_lag, f.gcc_eps = gccphat_out.data_ptr(), int(max_lag), float(gcc_eps)
        return dict(f=f, ref=ctypes.byref(f), keep=(logmel_out, mel_start, mel_w, gccphat_out))

    def observe_prepared_features(self, prep, spectrogram_ptr, audiogoal_ptr, stream: int, features) -> None:
        """``observe_prepared`` + the step's log-mel / GCC-PHAT on the same stream (``ss_ctx_observe_features``)."""
        rc = self.lib.ss_ctx_observe_features(self._h, prep["ref"], prep["n"], audiogoal_ptr, spectrogram_ptr, features["ref"],
                                              stream)
        if rc != 0:
            _lib.check(rc, "ss_ctx_observe_features")

    def bind_sims(self, state, index, has_distractor: bool = False):
        """Pointers to the int64 state columns (``ss_amd.vector.VectorSimState``) and the RIR index tables, for
        ``observe_sims``: built once, rebuilt by the caller when the index tables are rebuilt (``index.version``)."""
        flat, off, dim = index.tables()
        c = _lib.SsSimColumns()
        for name, arr in (("sound", state.sound), ("audio_index", state.audio_index), ("step_count", state.step_count),
                          ("duration", state.duration), ("recv", state.recv), ("src", state.src), ("rot", state.rot),
                          ("scene", state.scene)):
            assert arr.dtype == np.int64 and arr.flags.c_contiguous
            setattr(c, name, arr.ctypes.data)
        if has_distractor:
            c.dis_sound, c.dis_src = state.dis_sound.ctypes.data, state.dis_src.ctypes.data
        c.index_flat, c.index_off, c.index_dim = flat.ctypes.data, off.ctypes.data, dim.ctypes.data
        c.n_scenes, c.azimuths = int(dim.shape[0]), int(index.azimuths)
        n = int(state.sound.shape[0])
        miss, n_miss = np.zeros((max(n, 1),), np.int32), ctypes.c_int(0)
        return dict(cols=c, n=n, keep=(state, flat, off, dim), miss=miss, n_miss=n_miss,
                    c_args=(ctypes.byref(c), n), c_miss=(miss.ctypes.data, ctypes.addressof(n_miss)))   # built once: ctypes
                                                                                                        # conversions cost ~1 us each

    def sims_units(self, bound):
        """Host only: the unit columns ``observe_sims`` would render (and the same audio_index advance) -> (dict of
        int32 columns, missing env indices)."""
        n = bound["n"]
        out = np.zeros((5, max(n, 1)), np.int32)
        _lib.check(self.lib.ss_ctx_sims_units(self._h, ctypes.byref(bound["cols"]), n, out.ctypes.data,
                                              bound["miss"].ctypes.data, ctypes.addressof(bound["n_miss"])), "ss_ctx_sims_units")
        cols = dict(zip(("sound", "t0", "rir", "dis_sound", "dis_rir"), out[:, :n]))
        return cols, bound["miss"][:min(bound["n_miss"].value, n)].copy()

    def observe_sims(self, bound, spectrogram_out=None, audiogoal_out=None, stream: Optional[int] = None) -> np.ndarray:
        """One step straight from the bound state columns (ss_ctx_observe_sims: the whole per-step host work in C++).
        Returns the env indices whose RIR pair is not resident (then nothing was launched or advanced), else empty."""
        import torch
        sg = ag = None
        dev = None
        if spectrogram_out is not None:
            sg, dev = spectrogram_out.data_ptr(), spectrogram_out.device
        if audiogoal_out is not None:
            ag, dev = audiogoal_out.data_ptr(), audiogoal_out.device
        if dev is None:
            raise ValueError("observe_sims: pass spectrogram_out and / or audiogoal_out")
        if bound.get("dev") != dev:                    # shapes / dtypes checked once per (bound, device)
            n = bound["n"]
            for t, shape in ((spectrogram_out, (n,) + self.spectrogram_shape), (audiogoal_out, (n, 2, self.sr))):
                if t is not None:
                    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape
            bound["dev"] = dev
        if stream is None:
            stream = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
        if torch.cuda.current_device() == dev.index:
            rc = self.lib.ss_ctx_observe_sims(self._h, *bound["c_args"], ag, sg, *bound["c_miss"], stream)
        else:
            with torch.cuda.device(dev):
                rc = self.lib.ss_ctx_observe_sims(self._h, *bound["c_args"], ag, sg, *bound["c_miss"], stream)
        if rc != 0:
            _lib.check(rc, "ss_ctx_observe_sims")
        k = bound["n_miss"].value
        return bound["miss"][:min(k, bound["n"])] if k else _NO_MISS

    # ---- request records of a multi-process vector env (ss_amd/deferred.py) ------------------------------------------
    @staticmethod
    def request_tables(sound_keys, sound_ids, table_keys, table_ids, pair_keys, pair_slots, stale=None, last_used=None):
        """The sorted lookup tables of ``ss_ctx_observe_requests`` as its C struct (int64 numpy arrays, borrowed: the
        returned dict keeps them alive; rebuild it whenever one of them is replaced)."""
        arrs = [np.ascontiguousarray(a, np.int64) for a in (sound_keys, sound_ids, table_keys, table_ids, pair_keys, pair_slots)]
        t = _lib.SsRequestTables()
        for name, a in zip(("sound_keys", "sound_ids", "table_keys", "table_ids", "pair_keys", "pair_slots"), arrs):
            setattr(t, name, a.ctypes.data if a.shape[0] else None)
        t.n_sounds, t.n_tables, t.n_pairs = int(arrs[0].shape[0]), int(arrs[2].shape[0]), int(arrs[4].shape[0])
        if stale is not None:
            assert stale.dtype in (np.bool_, np.uint8) and stale.flags.c_contiguous
            t.stale, t.n_slots = stale.ctypes.data, int(stale.shape[0])
        if last_used is not None:                             # the store's LRU clock: last_used[slot] = t.tick per used slot
            assert last_used.dtype == np.int64 and last_used.flags.c_contiguous
            assert stale is None or stale.shape[0] == last_used.shape[0]
            t.last_used, t.n_slots = last_used.ctypes.data, int(last_used.shape[0])
        return dict(t=t, ref=ctypes.byref(t), keep=(arrs, stale, last_used))

    def observe_requests(self, recs: bytes, n: int, tables, spectrogram_ptr, audiogoal_ptr, stream: int, miss) -> int:
        """One step from the concatenated request records (n x SS_REQ_WORDS int64, as bytes): lookups, planning and the
        launch in ONE C call.  `miss` = dict(buf=int32[n] array, n=ctypes.c_int) reused across calls.  Returns the number
        of requests that could not be resolved (then nothing was launched; their indices are in miss['buf'])."""
        rc = self.lib.ss_ctx_observe_requests(self._h, recs, n, tables["ref"], audiogoal_ptr, spectrogram_ptr,
                                              miss["ptr"], miss["n_ptr"], stream)
        if rc != 0:
            _lib.check(rc, "ss_ctx_observe_requests")
        return miss["n"].value

    def observe_requests_load(self, recs: bytes, n: int, tables, loader, spectrogram_ptr, audiogoal_ptr, stream: int, miss) -> int:
        """``observe_requests`` with the miss path inside the call (``ss_ctx_observe_requests_load``): `loader` = the dict of
        ``renderer.RirStore.miss_loader`` (struct ss_miss_loader + what keeps its arrays alive).  Poses that are not resident are
        read, scattered and booked by the library when the fast path covers them (loader['s'].n_loaded says how many); otherwise
        the call behaves exactly like ``observe_requests``."""
        rc = self.lib.ss_ctx_observe_requests_load(self._h, recs, n, tables["ref"], loader["ref"], audiogoal_ptr, spectrogram_ptr,
                                                   miss["ptr"], miss["n_ptr"], stream)
        if rc != 0:
            _lib.check(rc, "ss_ctx_observe_requests_load")
        return miss["n"].value

    def requests_units(self, recs: bytes, n: int, tables):
        """Host only: the unit columns ``observe_requests`` would render -> (dict of int32 columns, missing request indices)."""
        out = np.zeros((5, max(n, 1)), np.int32)
        miss, n_miss = np.zeros((max(n, 1),), np.int32), ctypes.c_int(0)
        _lib.check(self.lib.ss_ctx_requests_units(self._h, recs, n, tables["ref"], out.ctypes.data, miss.ctypes.data,
                                                  ctypes.addressof(n_miss)), "ss_ctx_requests_units")
        return dict(zip(("sound", "t0", "rir", "dis_sound", "dis_rir"), out[:, :n])), miss[:min(n_miss.value, n)].copy()

    def plan(self, sound, t0, rir, dis_sound=None, dis_rir=None, last_rir=None, wrap=None, last_wrap=None):
        """The planner alone (host only): -> (unit descriptors int32 [n,8], launch flags, new windows int32 [w,5])."""
        u, n, keep = self._units(sound, t0, rir, dis_sound, dis_rir, last_rir, wrap, last_wrap)
        desc = np.zeros((n, 8), np.int32)
        flags, nw = ctypes.c_int(0), ctypes.c_int(0)
        cap = 2 * self.stats()["slots_per_key"] * n + 16       # at most two keys (terms) per unit, slots_per_key windows each
        wins = np.zeros((cap, 5), np.int32)
        _lib.check(self.lib.ss_ctx_plan(self._h, ctypes.byref(u), n, desc.ctypes.data, ctypes.addressof(flags),
                                        ctypes.addressof(nw), wins.ctypes.data, cap), "ss_ctx_plan")
        assert nw.value <= cap
        return desc, flags.value, wins[:nw.value]

    def stats(self) -> Dict[str, int]:
        out = np.zeros(8, np.int64)
        _lib.check(self.lib.ss_ctx_stats(self._h, out.ctypes.data), "ss_ctx_stats")
        keys = ("hits", "misses", "evictions", "grows", "capacity", "resident", "slots_per_key", "steps")
        return dict(zip(keys, (int(v) for v in out)))

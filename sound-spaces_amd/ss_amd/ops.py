"""Tensor-level entry points of the HIP audio path + their registration as PyTorch custom ops
(``torch.ops.ss_hip.*``).  PyTorch is plumbing here: device memory, streams, dispatch.  Every op runs the
hand-written gfx950 kernels of csrc/ through the C ABI (include/ss_hip.h) on the caller's current stream;
nothing syncs the host.

Reference call sites replaced (facebookresearch/sound-spaces):
  fftconv_binaural  soundspaces/simulator.py:629-647,649-664  (scipy.signal.fftconvolve per ear + slicing)
  spectrogram       soundspaces/tasks/nav.py:86-100           (librosa.stft -> abs -> 4x4 mean -> log1p)
  audio_obs         soundspaces/simulator.py:690-701          (both, fused; cache-miss path)
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .planning import KB, SPEC_FLOATS, ceil_div, spectrogram_shape

PAD_REFLECT, PAD_CONSTANT = 0, 1
FLAG_NO_DISTRACTOR = 1      # SS_FLAG_NO_DISTRACTOR: every unit descriptor has term 1 absent
FLAG_CROSSFADE = 2          # SS_FLAG_CROSSFADE: term 1 = previous step's RIR, blended over the first int(0.05*sr)+1 samples
FLAG_FIRST_BUCKET = 4       # SS_FLAG_FIRST_BUCKET: every bank index of the launch lies in bucket 0 of a bucketed bank

_PAD = {"reflect": PAD_REFLECT, "constant": PAD_CONSTANT, 0: 0, 1: 1}


def _stream(t: "torch.Tensor" = None) -> int:
    """Raw handle of the current stream of ``t``'s device (of the current device without ``t``).  Not
    ``torch.cuda.current_stream(None).cuda_stream``: that is ~10 us of device-index plumbing and object construction per call,
    a tenth of an eager (batch-1) observation."""
    idx = t.device.index if (t is not None and t.device.index is not None) else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.SsHipError(f"{name} must live on the GPU (got {t.device}); this path has no CPU implementation")
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def _bank_strides(rir_bank: torch.Tensor, interleaved: bool):
    if rir_bank.dim() != 3:
        raise ValueError("rir_bank must be [R,2,L] (planar) or [R,L,2] (wav-interleaved)")
    if interleaved:
        R, cap, two = rir_bank.shape
        assert two == 2
        return 2 * cap, 1, 2, cap
    R, two, cap = rir_bank.shape
    assert two == 2
    return 2 * cap, cap, 1, cap


def init() -> None:
    """Build the per-device constant tables (idempotent); also the explicit 'is the HIP path alive' probe."""
    _lib.check(_lib.load().ss_init(), "ss_init")


def source_windows_into(src: torch.Tensor, win_desc: torch.Tensor, spec_out: torch.Tensor) -> None:
    _chk(src, torch.float32, "src"); _chk(win_desc, torch.int32, "win_desc"); _chk(spec_out, torch.float32, "spec_out")
    W = win_desc.shape[0]
    assert win_desc.shape == (W, 4) and spec_out.numel() >= W * SPEC_FLOATS
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().ss_source_windows_f32(src.data_ptr(), win_desc.data_ptr(), spec_out.data_ptr(), W,
                                                     _stream(src)), "ss_source_windows_f32")


def source_windows(src: torch.Tensor, win_desc: torch.Tensor) -> torch.Tensor:
    out = torch.empty((win_desc.shape[0], SPEC_FLOATS), dtype=torch.float32, device=src.device)
    source_windows_into(src, win_desc, out)
    return out


def fftconv_binaural_into(spec, rir_bank, rir_len, unit_desc, out, n_valid: int, interleaved: bool = False,
                          flags: int = 0) -> None:
    _chk(spec, torch.float32, "spec"); _chk(rir_bank, torch.float32, "rir_bank"); _chk(rir_len, torch.int32, "rir_len")
    _chk(unit_desc, torch.int32, "unit_desc"); _chk(out, torch.float32, "out")
    N, two, out_len = out.shape
    assert two == 2 and unit_desc.shape == (N, 8)
    us, cs, es, cap = _bank_strides(rir_bank, interleaved)
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().ss_fftconv_binaural_f32(spec.data_ptr(), rir_bank.data_ptr(), rir_len.data_ptr(),
                                                       unit_desc.data_ptr(), out.data_ptr(), N, us, cs, es, cap,
                                                       n_valid, out_len, flags, _stream(spec)), "ss_fftconv_binaural_f32")


def fftconv_binaural(spec, rir_bank, rir_len, unit_desc, n_valid: int, out_len: int, interleaved: bool = False,
                     flags: int = 0):
    out = torch.empty((unit_desc.shape[0], 2, out_len), dtype=torch.float32, device=spec.device)
    fftconv_binaural_into(spec, rir_bank, rir_len, unit_desc, out, n_valid, interleaved, flags)
    return out


def spectrogram_into(x: torch.Tensor, out: torch.Tensor, pad_mode="reflect") -> None:
    _chk(x, torch.float32, "x"); _chk(out, torch.float32, "out")
    N, two, n = x.shape
    assert two == 2 and tuple(out.shape) == (N,) + spectrogram_shape(n)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().ss_spectrogram_f32(x.data_ptr(), out.data_ptr(), N, n, _PAD[pad_mode], _stream(x)),
                   "ss_spectrogram_f32")


def spectrogram(x: torch.Tensor, pad_mode="reflect") -> torch.Tensor:
    out = torch.empty((x.shape[0],) + spectrogram_shape(x.shape[2]), dtype=torch.float32, device=x.device)
    spectrogram_into(x, out, pad_mode)
    return out


def logmel_into(x: torch.Tensor, out: torch.Tensor, mel_start: torch.Tensor, mel_w: torch.Tensor, eps: float = 1e-6,
                pad_mode="reflect") -> None:
    """EXTENSION (not in the reference): out[n, j, t, c] = log(sum_i mel_w[j, i] |STFT(x[n, c])[mel_start[j]+i, t]|^2 + eps).
    mel_start int32 [n_mels], mel_w float32 [n_mels, max_len] on the device (planning.mel_filterbank_sparse)."""
    _chk(x, torch.float32, "x"); _chk(out, torch.float32, "out"); _chk(mel_start, torch.int32, "mel_start")
    _chk(mel_w, torch.float32, "mel_w")
    N, two, n = x.shape
    n_mels, max_len = mel_w.shape
    assert two == 2 and mel_start.shape == (n_mels,) and tuple(out.shape) == (N, n_mels, 1 + n // 160, 2)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().ss_logmel_f32(x.data_ptr(), out.data_ptr(), N, n, _PAD[pad_mode], mel_start.data_ptr(),
                                             mel_w.data_ptr(), n_mels, max_len, float(eps), _stream(x)), "ss_logmel_f32")


def logmel(x: torch.Tensor, mel_start: torch.Tensor, mel_w: torch.Tensor, eps: float = 1e-6, pad_mode="reflect"):
    out = torch.empty((x.shape[0], mel_w.shape[0], 1 + x.shape[2] // 160, 2), dtype=torch.float32, device=x.device)
    logmel_into(x, out, mel_start, mel_w, eps, pad_mode)
    return out


def gccphat_into(x: torch.Tensor, out: torch.Tensor, max_lag: int = 32, eps: float = 1e-8, pad_mode="reflect") -> None:
    """EXTENSION (not in the reference): per-frame GCC-PHAT between the two ears,
    out[n, i, t] = irfft(G / (|G| + eps))[(i - max_lag) mod 512], G = STFT(x[n, 0]) conj(STFT(x[n, 1]))."""
    _chk(x, torch.float32, "x"); _chk(out, torch.float32, "out")
    N, two, n = x.shape
    assert two == 2 and tuple(out.shape) == (N, 2 * max_lag + 1, 1 + n // 160)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().ss_gccphat_f32(x.data_ptr(), out.data_ptr(), N, n, _PAD[pad_mode], int(max_lag),
                                              float(eps), _stream(x)), "ss_gccphat_f32")


def gccphat(x: torch.Tensor, max_lag: int = 32, eps: float = 1e-8, pad_mode="reflect") -> torch.Tensor:
    out = torch.empty((x.shape[0], 2 * max_lag + 1, 1 + x.shape[2] // 160), dtype=torch.float32, device=x.device)
    gccphat_into(x, out, max_lag, eps, pad_mode)
    return out


def audio_features_into(x: torch.Tensor, spectrogram_out=None, logmel_out=None, gccphat_out=None, mel_start=None, mel_w=None,
                        mel_eps: float = 1e-6, max_lag: int = 32, gcc_eps: float = 1e-8, pad_mode="reflect") -> None:
    """EXTENSION: every STFT-derived feature from ONE pass over the waveform x [N, 2, n] (``ss_audio_features_f32``): any
    subset of the pooled spectrogram [N,65,T4,2] (nav.py:86-100), log-mel [N,n_mels,T,2] and GCC-PHAT [N,2*max_lag+1,T];
    same results as ``spectrogram_into`` / ``logmel_into`` / ``gccphat_into``, which each re-read x and redo the STFT."""
    _chk(x, torch.float32, "x")
    N, two, n = x.shape
    assert two == 2 and (spectrogram_out is not None or logmel_out is not None or gccphat_out is not None)
    n_mels = max_len = 0
    if spectrogram_out is not None:
        _chk(spectrogram_out, torch.float32, "spectrogram_out")
        assert tuple(spectrogram_out.shape) == (N,) + spectrogram_shape(n)
    if logmel_out is not None:
        _chk(logmel_out, torch.float32, "logmel_out"); _chk(mel_start, torch.int32, "mel_start"); _chk(mel_w, torch.float32, "mel_w")
        n_mels, max_len = mel_w.shape
        assert mel_start.shape == (n_mels,) and tuple(logmel_out.shape) == (N, n_mels, 1 + n // 160, 2)
    if gccphat_out is not None:
        _chk(gccphat_out, torch.float32, "gccphat_out")
        assert tuple(gccphat_out.shape) == (N, 2 * max_lag + 1, 1 + n // 160)
    ptr = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().ss_audio_features_f32(x.data_ptr(), N, n, _PAD[pad_mode], ptr(spectrogram_out), ptr(logmel_out),
                                                     ptr(mel_start), ptr(mel_w), int(n_mels), int(max_len), float(mel_eps),
                                                     ptr(gccphat_out), int(max_lag), float(gcc_eps), _stream(x)),
                   "ss_audio_features_f32")


def audio_features(x: torch.Tensor, want=("logmel", "gccphat"), mel_start=None, mel_w=None, mel_eps: float = 1e-6,
                   max_lag: int = 32, gcc_eps: float = 1e-8, pad_mode="reflect"):
    """-> dict of the wanted features ("spectrogram", "logmel", "gccphat"), one launch."""
    N, _, n = x.shape
    out = {}
    if "spectrogram" in want:
        out["spectrogram"] = torch.empty((N,) + spectrogram_shape(n), dtype=torch.float32, device=x.device)
    if "logmel" in want:
        out["logmel"] = torch.empty((N, mel_w.shape[0], 1 + n // 160, 2), dtype=torch.float32, device=x.device)
    if "gccphat" in want:
        out["gccphat"] = torch.empty((N, 2 * max_lag + 1, 1 + n // 160), dtype=torch.float32, device=x.device)
    audio_features_into(x, out.get("spectrogram"), out.get("logmel"), out.get("gccphat"), mel_start, mel_w, mel_eps, max_lag,
                        gcc_eps, pad_mode)
    return out


def audio_obs_into(spec, rir_bank, rir_len, unit_desc, audiogoal, spectrogram_out, n_valid: int, out_len: int,
                   pad_mode="reflect", interleaved: bool = False, flags: int = 0) -> None:
    """Fused observation.  ``audiogoal`` may be None (the waveform then never leaves the CU).  Cross-faded rows longer
    than one partition block run as two launches when a buffer is given (faster), as one without."""
    _chk(spec, torch.float32, "spec"); _chk(rir_bank, torch.float32, "rir_bank"); _chk(rir_len, torch.int32, "rir_len")
    _chk(unit_desc, torch.int32, "unit_desc"); _chk(spectrogram_out, torch.float32, "spectrogram_out")
    N = unit_desc.shape[0]
    assert tuple(spectrogram_out.shape) == (N,) + spectrogram_shape(out_len)
    ag_ptr = None
    if audiogoal is not None:
        _chk(audiogoal, torch.float32, "audiogoal")
        assert tuple(audiogoal.shape) == (N, 2, out_len)
        ag_ptr = audiogoal.data_ptr()
    us, cs, es, cap = _bank_strides(rir_bank, interleaved)
    with torch.cuda.device(spec.device):
        _lib.check(_lib.load().ss_audio_obs_f32(spec.data_ptr(), rir_bank.data_ptr(), rir_len.data_ptr(),
                                                unit_desc.data_ptr(), ag_ptr, spectrogram_out.data_ptr(), N, us, cs,
                                                es, cap, n_valid, out_len, _PAD[pad_mode], flags, _stream(spec)),
                   "ss_audio_obs_f32")


def audio_obs(spec, rir_bank, rir_len, unit_desc, n_valid: int, out_len: int, pad_mode="reflect",
              want_audiogoal: bool = False, interleaved: bool = False, flags: int = 0):
    N = unit_desc.shape[0]
    from .planning import wide_one_block
    need_ag = want_audiogoal or (out_len > KB and not wide_one_block(out_len, n_valid)
                                 and (bool(flags & FLAG_CROSSFADE) or out_len > 3 * KB))
    ag = torch.empty((N, 2, out_len), dtype=torch.float32, device=spec.device) if need_ag else None
    sg = torch.empty((N,) + spectrogram_shape(out_len), dtype=torch.float32, device=spec.device)
    audio_obs_into(spec, rir_bank, rir_len, unit_desc, ag, sg, n_valid, out_len, pad_mode, interleaved, flags)
    return ag, sg


# ---- spectral RIR bank ------------------------------------------------------------------------------------------
def rir_spectra_into(rir_bank: torch.Tensor, hspec: torch.Tensor, first: int = 0, count: Optional[int] = None) -> None:
    """Block spectra of bank entries [first, first+count) of a planar bank [R,2,cap] into hspec [R,2,hb,SPEC_FLOATS]
    (ss_rir_spectra_f32; one-off work at bank load, synchronous)."""
    _chk(rir_bank, torch.float32, "rir_bank"); _chk(hspec, torch.float32, "hspec")
    R, two, cap = rir_bank.shape
    hb = ceil_div(cap, KB)
    assert two == 2 and tuple(hspec.shape) == (R, 2, hb, SPEC_FLOATS)
    count = R - first if count is None else count
    assert 0 <= first and first + count <= R
    with torch.cuda.device(rir_bank.device):
        _lib.check(_lib.load().ss_rir_spectra_f32(rir_bank[first:].data_ptr(), hspec[first:].data_ptr(), count, 2 * cap, cap,
                                                  cap, _stream(rir_bank)), "ss_rir_spectra_f32")


def rir_spectra(rir_bank: torch.Tensor) -> torch.Tensor:
    R, _, cap = rir_bank.shape
    hspec = torch.empty((R, 2, ceil_div(cap, KB), SPEC_FLOATS), dtype=torch.float32, device=rir_bank.device)
    rir_spectra_into(rir_bank, hspec)
    return hspec


def fftconv_binaural_spec_into(spec, hspec, rir_len, unit_desc, out, n_valid: int, flags: int = 0) -> None:
    _chk(spec, torch.float32, "spec"); _chk(hspec, torch.float32, "hspec"); _chk(rir_len, torch.int32, "rir_len")
    _chk(unit_desc, torch.int32, "unit_desc"); _chk(out, torch.float32, "out")
    N, two, out_len = out.shape
    assert two == 2 and unit_desc.shape == (N, 8) and hspec.dim() == 4
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().ss_fftconv_binaural_spec_f32(spec.data_ptr(), hspec.data_ptr(), rir_len.data_ptr(),
                                                            unit_desc.data_ptr(), out.data_ptr(), N, hspec.shape[2],
                                                            n_valid, out_len, flags, _stream(spec)),
                   "ss_fftconv_binaural_spec_f32")


def audio_obs_spec_into(spec, hspec, rir_len, unit_desc, audiogoal, spectrogram_out, n_valid: int, out_len: int,
                        pad_mode="reflect", flags: int = 0) -> None:
    _chk(spec, torch.float32, "spec"); _chk(hspec, torch.float32, "hspec"); _chk(rir_len, torch.int32, "rir_len")
    _chk(unit_desc, torch.int32, "unit_desc"); _chk(spectrogram_out, torch.float32, "spectrogram_out")
    N = unit_desc.shape[0]
    assert tuple(spectrogram_out.shape) == (N,) + spectrogram_shape(out_len) and hspec.dim() == 4
    ag_ptr = None
    if audiogoal is not None:
        _chk(audiogoal, torch.float32, "audiogoal")
        assert tuple(audiogoal.shape) == (N, 2, out_len)
        ag_ptr = audiogoal.data_ptr()
    with torch.cuda.device(spec.device):
        _lib.check(_lib.load().ss_audio_obs_spec_f32(spec.data_ptr(), hspec.data_ptr(), rir_len.data_ptr(),
                                                     unit_desc.data_ptr(), ag_ptr, spectrogram_out.data_ptr(), N,
                                                     hspec.shape[2], n_valid, out_len, _PAD[pad_mode], flags, _stream(spec)),
                   "ss_audio_obs_spec_f32")


# ---- length-bucketed RIR bank ------------------------------------------------------------------------
def bucket_array(banks, firsts, spectral: bool):
    """ctypes array of ss_rir_bucket for per-bucket (data [n,2,cap], spectra or None) tensors; keep it alive with the
    tensors it points to."""
    arr = (_lib.SsRirBucket * len(banks))()
    for b, (bank, first) in enumerate(zip(banks, firsts)):
        _chk(bank.data, torch.float32, "bucket data")
        arr[b].rir = bank.data.data_ptr()
        arr[b].hspec = bank.spectra.data_ptr() if (spectral and bank.spectra is not None) else None
        arr[b].first, arr[b].n_entries, arr[b].cap, arr[b].reserved = int(first), int(bank.data.shape[0]), int(bank.data.shape[2]), 0
    return arr


def audio_obs_buckets_into(spec, buckets, n_buckets: int, rir_len, unit_desc, audiogoal, spectrogram_out, n_valid: int,
                           out_len: int, pad_mode="reflect", flags: int = 0) -> None:
    """``audio_obs_into`` on a length-bucketed bank: ``buckets`` = ``bucket_array(...)``, rir_len int32 [total slots]."""
    _chk(spec, torch.float32, "spec"); _chk(rir_len, torch.int32, "rir_len"); _chk(unit_desc, torch.int32, "unit_desc")
    N = unit_desc.shape[0]
    ag_ptr = sg_ptr = None
    if audiogoal is not None:
        _chk(audiogoal, torch.float32, "audiogoal")
        assert tuple(audiogoal.shape) == (N, 2, out_len)
        ag_ptr = audiogoal.data_ptr()
    with torch.cuda.device(spec.device):
        if spectrogram_out is None:
            _lib.check(_lib.load().ss_fftconv_binaural_buckets_f32(spec.data_ptr(), ctypes_ref(buckets), n_buckets,
                                                                   rir_len.data_ptr(), unit_desc.data_ptr(), ag_ptr, N, n_valid,
                                                                   out_len, flags, _stream(spec)), "ss_fftconv_binaural_buckets_f32")
            return
        _chk(spectrogram_out, torch.float32, "spectrogram_out")
        assert tuple(spectrogram_out.shape) == (N,) + spectrogram_shape(out_len)
        sg_ptr = spectrogram_out.data_ptr()
        _lib.check(_lib.load().ss_audio_obs_buckets_f32(spec.data_ptr(), ctypes_ref(buckets), n_buckets, rir_len.data_ptr(),
                                                        unit_desc.data_ptr(), ag_ptr, sg_ptr, N, n_valid, out_len,
                                                        _PAD[pad_mode], flags, _stream(spec)), "ss_audio_obs_buckets_f32")


def ctypes_ref(arr):
    import ctypes
    return ctypes.cast(arr, ctypes.c_void_p)


def intensity(audiogoal: torch.Tensor, num_frame: int = 150) -> torch.Tensor:
    """av_wan Intensity (ss_baselines/av_wan/avwan_sensors.py:91-100): [N,2,T] -> [N] mean-square of the 150 samples
    after the onset."""
    _chk(audiogoal, torch.float32, "audiogoal")
    N, two, n = audiogoal.shape
    assert two == 2
    out = torch.empty((N,), dtype=torch.float32, device=audiogoal.device)
    with torch.cuda.device(audiogoal.device):
        _lib.check(_lib.load().ss_intensity_f32(audiogoal.data_ptr(), out.data_ptr(), N, n, num_frame, _stream(audiogoal)),
                   "ss_intensity_f32")
    return out


# ---- torch.ops.ss_hip.* ------------------------------------------------------------------------------
def _load_native_ops() -> bool:
    """The TORCH_LIBRARY extension (csrc/ss_torch_ops.cpp, built in-tree by build.py): the hot ops - spectrogram, audio_obs,
    ctx_observe and the one-dispatch eager observation eager_obs - are C++ dispatches straight into libss_hip.so.  Without
    the file (a tree that was never built) the same schemas are registered from Python below, over ctypes."""
    import logging
    import os
    so = os.path.join(os.path.dirname(_lib.SO_PATH), "libss_torch_ops.so")
    if not os.path.exists(so):
        return False
    try:
        torch.ops.load_library(so)
        return True
    except OSError as e:                                    # e.g. built against another torch: keep the Python registrations
        logging.warning("ss_amd: %s did not load (%s); torch.ops.ss_hip.* stay on the Python registrations", so, e)
        return False


NATIVE_OPS = _load_native_ops()
# 2: the extension also defines source_windows / fftconv_binaural / rir_spectra / *_spec / audio_features / intensity
NATIVE_LEVEL = int(torch.ops.ss_hip.native_ops()) if NATIVE_OPS else 0


def _register():
    lib = torch.library.Library("ss_hip", "FRAGMENT" if NATIVE_OPS else "DEF")
    py2 = NATIVE_LEVEL < 2                                  # these ops are NOT defined by the C++ extension: register them here
    if py2:
        lib.define("source_windows(Tensor src, Tensor win_desc) -> Tensor")
        lib.define("fftconv_binaural(Tensor spec, Tensor rir_bank, Tensor rir_len, Tensor unit_desc, int n_valid, "
                   "int out_len, bool interleaved=False, int flags=0) -> Tensor")
    if not NATIVE_OPS:
        lib.define("spectrogram(Tensor x, int pad_mode=0) -> Tensor")
        lib.define("audio_obs(Tensor spec, Tensor rir_bank, Tensor rir_len, Tensor unit_desc, int n_valid, int out_len, "
                   "int pad_mode=0, bool interleaved=False, int flags=0) -> (Tensor, Tensor)")
    if py2:
        lib.define("intensity(Tensor audiogoal, int num_frame=150) -> Tensor")
    lib.define("gccphat(Tensor x, int max_lag=32, float eps=1e-8, int pad_mode=0) -> Tensor")
    lib.impl("gccphat", lambda x, max_lag=32, eps=1e-8, pad_mode=0: gccphat(x, max_lag, eps, pad_mode), "CUDA")
    lib.impl("gccphat", lambda x, max_lag=32, eps=1e-8, pad_mode=0:
             x.new_empty((x.shape[0], 2 * max_lag + 1, 1 + x.shape[2] // 160)), "Meta")
    lib.define("logmel(Tensor x, Tensor mel_start, Tensor mel_w, float eps=1e-6, int pad_mode=0) -> Tensor")
    lib.impl("logmel", lambda x, ms, mw, eps=1e-6, pad_mode=0: logmel(x, ms, mw, eps, pad_mode), "CUDA")
    lib.impl("logmel", lambda x, ms, mw, eps=1e-6, pad_mode=0:
             x.new_empty((x.shape[0], mw.shape[0], 1 + x.shape[2] // 160, 2)), "Meta")
    if py2:
        lib.define("audio_features(Tensor x, Tensor mel_start, Tensor mel_w, float mel_eps=1e-6, int max_lag=32, float gcc_eps=1e-8, "
                   "int pad_mode=0) -> (Tensor, Tensor)")

    def _audio_features(x, ms, mw, mel_eps=1e-6, max_lag=32, gcc_eps=1e-8, pad_mode=0):
        o = audio_features(x, ("logmel", "gccphat"), ms, mw, mel_eps, max_lag, gcc_eps, pad_mode)
        return o["logmel"], o["gccphat"]
    if py2:
        lib.impl("audio_features", _audio_features, "CUDA")
    lib.impl("audio_features", lambda x, ms, mw, mel_eps=1e-6, max_lag=32, gcc_eps=1e-8, pad_mode=0:
             (x.new_empty((x.shape[0], mw.shape[0], 1 + x.shape[2] // 160, 2)),
              x.new_empty((x.shape[0], 2 * max_lag + 1, 1 + x.shape[2] // 160))), "Meta")
    if py2:
        lib.impl("intensity", intensity, "CUDA")
        lib.impl("source_windows", source_windows, "CUDA")
        lib.impl("fftconv_binaural", fftconv_binaural, "CUDA")
    lib.impl("intensity", lambda a, num_frame=150: a.new_empty((a.shape[0],)), "Meta")
    def _audio_obs(spec, rir_bank, rir_len, unit_desc, n_valid, out_len, pad_mode=0, interleaved=False, flags=0):
        ag, sg = audio_obs(spec, rir_bank, rir_len, unit_desc, n_valid, out_len, pad_mode, True, interleaved, flags)
        return ag, sg
    if not NATIVE_OPS:
        lib.impl("spectrogram", lambda x, pad_mode=0: spectrogram(x, pad_mode), "CUDA")
        lib.impl("audio_obs", _audio_obs, "CUDA")

    # spectral RIR bank: the bank builder + the two *_spec entry points
    if py2:
        lib.define("rir_spectra(Tensor rir_bank) -> Tensor")
        lib.impl("rir_spectra", rir_spectra, "CUDA")
    lib.impl("rir_spectra", lambda b: b.new_empty((b.shape[0], 2, ceil_div(b.shape[2], KB), SPEC_FLOATS)), "Meta")
    if py2:
        lib.define("fftconv_binaural_spec(Tensor spec, Tensor hspec, Tensor rir_len, Tensor unit_desc, int n_valid, int out_len, "
                   "int flags=0) -> Tensor")

    def _conv_spec(spec, hspec, rir_len, unit_desc, n_valid, out_len, flags=0):
        out = torch.empty((unit_desc.shape[0], 2, out_len), dtype=torch.float32, device=spec.device)
        fftconv_binaural_spec_into(spec, hspec, rir_len, unit_desc, out, n_valid, flags)
        return out
    if py2:
        lib.impl("fftconv_binaural_spec", _conv_spec, "CUDA")
    lib.impl("fftconv_binaural_spec", lambda spec, h, l, d, n_valid, out_len, flags=0: spec.new_empty((d.shape[0], 2, out_len)),
             "Meta")
    if py2:
        lib.define("audio_obs_spec(Tensor spec, Tensor hspec, Tensor rir_len, Tensor unit_desc, int n_valid, int out_len, "
                   "int pad_mode=0, int flags=0) -> (Tensor, Tensor)")

    def _obs_spec(spec, hspec, rir_len, unit_desc, n_valid, out_len, pad_mode=0, flags=0):
        N = unit_desc.shape[0]
        ag = torch.empty((N, 2, out_len), dtype=torch.float32, device=spec.device)
        sg = torch.empty((N,) + spectrogram_shape(out_len), dtype=torch.float32, device=spec.device)
        audio_obs_spec_into(spec, hspec, rir_len, unit_desc, ag, sg, n_valid, out_len, pad_mode, flags)
        return ag, sg
    if py2:
        lib.impl("audio_obs_spec", _obs_spec, "CUDA")
    lib.impl("audio_obs_spec", lambda spec, h, l, d, n_valid, out_len, pad_mode=0, flags=0:
             (spec.new_empty((d.shape[0], 2, out_len)), spec.new_empty((d.shape[0],) + spectrogram_shape(out_len))), "Meta")
    # the observe-level op: one vector step through a context (planner + window cache + descriptor ring in the library).
    # `ctx` = AudioContext.handle (an integer registered by ss_amd.context); unit columns are int32 CPU tensors; the
    # spectrogram rows are written in place (rollout rows) and returned.
    def _ctx_observe(ctx, sound, t0, rir, spectrogram):
        from .context import AudioContext
        AudioContext.from_handle(ctx).observe(sound.numpy(), t0.numpy(), rir.numpy(), spectrogram_out=spectrogram)
        return spectrogram
    if not NATIVE_OPS:
        lib.define("ctx_observe(int ctx, Tensor sound, Tensor t0, Tensor rir, Tensor(a!) spectrogram) -> Tensor(a!)")
        lib.impl("ctx_observe", _ctx_observe, "CompositeExplicitAutograd")

    # shape functions (Meta) so the ops compose with tracing / fake tensors
    lib.impl("source_windows", lambda src, wd: src.new_empty((wd.shape[0], SPEC_FLOATS)), "Meta")
    lib.impl("fftconv_binaural", lambda spec, b, l, d, n_valid, out_len, interleaved=False, flags=0:
             spec.new_empty((d.shape[0], 2, out_len)), "Meta")
    lib.impl("spectrogram", lambda x, pad_mode=0: x.new_empty((x.shape[0],) + spectrogram_shape(x.shape[2])), "Meta")
    lib.impl("audio_obs", lambda spec, b, l, d, n_valid, out_len, pad_mode=0, interleaved=False, flags=0:
             (spec.new_empty((d.shape[0], 2, out_len)), spec.new_empty((d.shape[0],) + spectrogram_shape(out_len))),
             "Meta")
    return lib


_TORCH_LIB = _register()

"""ctypes binding of libss_hip.so (include/ss_hip.h).  There is NO fallback: if the HIP library is
missing or a call fails, the error is raised — the product never computes this path on the CPU."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.normpath(os.path.join(_HERE, "..", "csrc", "libss_hip.so"))

EXPORTS = ("ss_block_len", "ss_spec_floats", "ss_version", "ss_init", "ss_source_windows_f32",
           "ss_fftconv_binaural_f32", "ss_spectrogram_f32", "ss_audio_obs_f32", "ss_intensity_f32", "ss_logmel_f32", "ss_gccphat_f32",
           "ss_ctx_create", "ss_ctx_destroy", "ss_ctx_add_source", "ss_ctx_add_source_len", "ss_ctx_set_rir_bank",
           "ss_ctx_observe", "ss_ctx_plan", "ss_ctx_stats", "ss_ctx_set_rir_spectra",
           "ss_rir_spectra_f32", "ss_fftconv_binaural_spec_f32", "ss_audio_obs_spec_f32", "ss_ctx_observe_sims",
           "ss_ctx_sims_units", "ss_ctx_set_overlap", "ss_ctx_join", "ss_fftconv_binaural_buckets_f32",
           "ss_audio_obs_buckets_f32", "ss_ctx_set_rir_buckets", "ss_release_scratch",
           "ss_source_windows32_f32", "ss_audio_obs32_f32", "ss_ctx_observe_requests", "ss_ctx_requests_units", "ss_audio_features_f32", "ss_ctx_observe_features",
           "ss_wav_read_rirs_f32", "ss_rows_gather_f32", "ss_bank_scatter_rows_f32", "ss_ctx_set_chip_share", "ss_ctx_set_spectral_policy", "ss_ctx_observe_requests_load", "ss_ctx_load_rir_files")


class SsRirBucket(ctypes.Structure):
    """struct ss_rir_bucket of include/ss_hip.h."""
    _fields_ = [("rir", ctypes.c_void_p), ("hspec", ctypes.c_void_p), ("first", ctypes.c_int), ("n_entries", ctypes.c_int),
                ("cap", ctypes.c_int), ("reserved", ctypes.c_int)]


class SsMissLoader(ctypes.Structure):
    """struct ss_miss_loader of include/ss_hip.h (ss_ctx_observe_requests_load: the miss path inside the call)."""
    _fields_ = [("table_dirs", ctypes.POINTER(ctypes.c_char_p)), ("n_table_dirs", ctypes.c_int),
                ("pair_keys", ctypes.c_void_p), ("pair_slots", ctypes.c_void_p), ("pair_cap", ctypes.c_int),
                ("free_slots", ctypes.c_void_p), ("n_free", ctypes.c_int),
                ("bank", ctypes.c_void_p), ("bank_unit_stride", ctypes.c_longlong), ("bank_chan_stride", ctypes.c_int),
                ("cap", ctypes.c_int), ("dev_len", ctypes.c_void_p), ("host_len", ctypes.c_void_p), ("clipped", ctypes.c_void_p),
                ("spec_stale", ctypes.c_void_p), ("keep", ctypes.c_int),
                ("stage", ctypes.c_void_p), ("stage_slot", ctypes.c_void_p), ("stage_len", ctypes.c_void_p),
                ("stage_desc", ctypes.c_void_p), ("stage_rows", ctypes.c_int), ("threads", ctypes.c_int),
                ("used", ctypes.c_void_p), ("use_seq", ctypes.c_void_p), ("evicted_slot", ctypes.c_void_p),
                ("n_evicted", ctypes.c_int), ("evict_cap", ctypes.c_int),
                ("loaded_key", ctypes.c_void_p), ("loaded_slot", ctypes.c_void_p), ("loaded_frames", ctypes.c_void_p),
                ("n_loaded", ctypes.c_int), ("loaded_cap", ctypes.c_int)]


class SsSimColumns(ctypes.Structure):
    """struct ss_sim_columns of include/ss_hip.h."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("sound", "audio_index", "step_count", "duration", "recv", "src", "rot", "scene",
                                              "dis_sound", "dis_src", "index_flat", "index_off", "index_dim")] + \
               [("n_scenes", ctypes.c_int), ("azimuths", ctypes.c_int)]


class SsRequestTables(ctypes.Structure):
    """struct ss_request_tables of include/ss_hip.h."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("sound_keys", "sound_ids", "table_keys", "table_ids", "pair_keys", "pair_slots",
                                              "stale")] + \
               [(n, ctypes.c_int) for n in ("n_sounds", "n_tables", "n_pairs", "n_slots")] + \
               [("last_used", ctypes.c_void_p), ("tick", ctypes.c_longlong)]


class SsFeatures(ctypes.Structure):
    """struct ss_features of include/ss_hip.h."""
    _fields_ = [("logmel", ctypes.c_void_p), ("mel_start", ctypes.c_void_p), ("mel_w", ctypes.c_void_p), ("n_mels", ctypes.c_int),
                ("max_len", ctypes.c_int), ("mel_eps", ctypes.c_float), ("gccphat", ctypes.c_void_p), ("max_lag", ctypes.c_int),
                ("gcc_eps", ctypes.c_float)]


class SsUnits(ctypes.Structure):
    """struct ss_units of include/ss_hip.h: host struct-of-arrays descriptors of one step."""
    _fields_ = [("sound", ctypes.c_void_p), ("t0", ctypes.c_void_p), ("rir", ctypes.c_void_p),
                ("dis_sound", ctypes.c_void_p), ("dis_rir", ctypes.c_void_p), ("last_rir", ctypes.c_void_p),
                ("wrap", ctypes.c_void_p), ("last_wrap", ctypes.c_void_p)]

_lib = None


class SsHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise SsHipError(f"{SO_PATH} not found: build it with `python sound-spaces_amd/build.py` "
                         "(or __graft_entry__.build()); there is no CPU fallback for this path")
    lib = ctypes.CDLL(SO_PATH)
    c_int, c_ll, vp = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p
    lib.ss_block_len.restype = c_int
    lib.ss_spec_floats.restype = c_int
    lib.ss_version.restype = c_int
    lib.ss_init.restype = c_int
    lib.ss_source_windows_f32.argtypes = [vp, vp, vp, c_int, vp]
    lib.ss_fftconv_binaural_f32.argtypes = [vp, vp, vp, vp, vp, c_int, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, vp]
    lib.ss_spectrogram_f32.argtypes = [vp, vp, c_int, c_int, c_int, vp]
    lib.ss_audio_obs_f32.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp]
    lib.ss_intensity_f32.argtypes = [vp, vp, c_int, c_int, c_int, vp]
    lib.ss_gccphat_f32.argtypes = [vp, vp, c_int, c_int, c_int, c_int, ctypes.c_float, vp]
    lib.ss_logmel_f32.argtypes = [vp, vp, c_int, c_int, c_int, vp, vp, c_int, c_int, ctypes.c_float, vp]
    pp = ctypes.POINTER(ctypes.c_void_p)
    lib.ss_ctx_create.argtypes = [pp, c_int, c_int, c_int, c_int, c_int]
    lib.ss_ctx_destroy.argtypes = [vp]
    lib.ss_ctx_add_source.argtypes = [vp, vp, c_int, c_int]
    lib.ss_ctx_add_source_len.argtypes = [vp, c_int]
    lib.ss_ctx_set_rir_bank.argtypes = [vp, vp, vp, c_ll, c_int, c_int, c_int]
    lib.ss_ctx_observe.argtypes = [vp, ctypes.POINTER(SsUnits), c_int, vp, vp, vp]
    lib.ss_ctx_plan.argtypes = [vp, ctypes.POINTER(SsUnits), c_int, vp, vp, vp, vp, c_int]
    lib.ss_ctx_stats.argtypes = [vp, vp]
    lib.ss_ctx_observe_sims.argtypes = [vp, ctypes.POINTER(SsSimColumns), c_int, vp, vp, vp, vp, vp]
    lib.ss_ctx_sims_units.argtypes = [vp, ctypes.POINTER(SsSimColumns), c_int, vp, vp, vp]
    lib.ss_ctx_set_rir_spectra.argtypes = [vp, vp, c_int]
    lib.ss_ctx_set_overlap.argtypes = [vp, c_int]
    lib.ss_fftconv_binaural_buckets_f32.argtypes = [vp, vp, c_int, vp, vp, vp, c_int, c_int, c_int, c_int, vp]
    lib.ss_audio_obs_buckets_f32.argtypes = [vp, vp, c_int, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp]
    lib.ss_ctx_set_rir_buckets.argtypes = [vp, vp, c_int, vp]
    lib.ss_ctx_join.argtypes = [vp, vp]
    lib.ss_ctx_set_chip_share.argtypes = [vp, c_int]
    lib.ss_ctx_set_spectral_policy.argtypes = [vp, c_int]
    lib.ss_rir_spectra_f32.argtypes = [vp, vp, c_int, c_ll, c_int, c_int, vp]
    lib.ss_fftconv_binaural_spec_f32.argtypes = [vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp]
    lib.ss_audio_obs_spec_f32.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp]
    lib.ss_source_windows32_f32.argtypes = [vp, vp, vp, c_int, vp]
    lib.ss_audio_obs32_f32.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, vp]
    lib.ss_audio_features_f32.argtypes = [vp, c_int, c_int, c_int, vp, vp, vp, vp, c_int, c_int, ctypes.c_float, vp, c_int,
                                          ctypes.c_float, vp]
    lib.ss_ctx_observe_features.argtypes = [vp, ctypes.POINTER(SsUnits), c_int, vp, vp, ctypes.POINTER(SsFeatures), vp]
    lib.ss_ctx_observe_requests.argtypes = [vp, vp, c_int, vp, vp, vp, vp, vp, vp]
    lib.ss_ctx_requests_units.argtypes = [vp, vp, c_int, vp, vp, vp, vp]
    lib.ss_ctx_observe_requests_load.argtypes = [vp, vp, c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.ss_ctx_load_rir_files.argtypes = [vp, vp, vp, c_int, vp, c_ll, c_int, vp]
    lib.ss_wav_read_rirs_f32.argtypes = [vp, c_int, vp, c_ll, c_int, c_int, c_int, vp, vp, vp, c_int]
    lib.ss_rows_gather_f32.argtypes = [vp, vp, c_int, vp, c_ll, c_int, c_int]
    lib.ss_bank_scatter_rows_f32.argtypes = [vp, c_ll, vp, vp, c_int, vp, c_ll, c_int, c_int, vp, vp]
    for name in EXPORTS:
        getattr(lib, name).restype = c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = "invalid argument" if rc == -1 else f"hipError_t {-rc}"
        raise SsHipError(f"{what} failed: {kind}")


WAV_OK, WAV_UNSUPPORTED, WAV_EMPTY, WAV_MISSING, WAV_TOO_LONG = 0, 1, 2, 3, 4


_n_threads = 0


def _default_threads() -> int:
    """threads of the library's host pool for this process (asked once: the affinity mask is a system call)"""
    global _n_threads
    if _n_threads <= 0:
        _n_threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    return _n_threads


def wav_read_rirs(paths, dst, cap: int, keep: int = -1, planar: bool = False, threads: int = 0):
    """ss_wav_read_rirs_f32: the float32 stereo wav files `paths` -> rows of the HOST float32 numpy array `dst`
    ([n, cap, 2] wav-interleaved, or [n, 2, cap] with planar=True; C-contiguous, typically the numpy view of a pinned
    torch tensor).  -> (kept, frames, status) int32 arrays (include/ss_hip.h).  Host-only: works without a GPU."""
    import numpy as np
    n = len(paths)
    assert dst.dtype == np.float32 and dst.flags.c_contiguous and dst.shape[0] >= n and dst[0].size == 2 * cap
    kept, frames, status = (np.zeros((n,), np.int32) for _ in range(3))
    if n == 0:
        return kept, frames, status
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    if threads <= 0:
        threads = _default_threads()
    check(load().ss_wav_read_rirs_f32(ctypes.cast(arr, ctypes.c_void_p), n, dst.ctypes.data, 2 * cap, cap, keep, int(planar),
                                      kept.ctypes.data, frames.ctypes.data, status.ctypes.data, threads), "ss_wav_read_rirs_f32")
    return kept, frames, status

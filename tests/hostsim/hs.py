"""TEST INFRASTRUCTURE: ctypes front-end of tests/hostsim/libss_hostsim.so (the HIP kernels compiled for
the host, see hostsim.cpp).  Mirrors the call sequence of the product renderer so descriptor planning is
exercised too."""
import ctypes
import os
import subprocess

import numpy as np

from ss_amd import planning as P

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "libss_hostsim.so")
        srcs = [os.path.join(HERE, f) for f in ("hostsim.cpp", "hip_shim.h")]
        csrc = os.path.join(HERE, "..", "..", "sound-spaces_amd", "csrc")
        srcs += [os.path.join(csrc, f) for f in ("ss_kernels.hpp", "ss_fft_core.hpp", "ss_tables.hpp", "ss_kernels32.hpp", "ss_fft_core32.hpp", "ss_features.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            cxx = os.environ.get("SS_HOSTSIM_CXX", "/opt/rocm/lib/llvm/bin/clang++")   # needs ext_vector_type
            subprocess.check_call([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                                   "-Wno-pass-failed", "-include", "hip_shim.h", "hostsim.cpp", "-o", so], cwd=HERE)
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def run(sources, rir_bank, rir_len, units, n_valid, out_len, fuse=False, want_spectrogram=False, pad_mode=0,
        interleaved=False, simple=True, persist=0, crossfade=False, spectral=False, row_wgs=0, want_audiogoal=True, row_stash=False,
        bucket2=None, core32=False, tab=False, parts_log2=0, row_blocks=False):
    """sources: list of f32 arrays; rir_bank f32 [R,2,cap] planar (zero padded); units: list of dicts
    {sound, t0, rir, wrap=False, dis_sound=None, dis_t0=0, dis_rir=-1} (rir < 0: silent); with crossfade=True a unit's
    {last_rir, last_wrap} is the previous step's RIR (term 1 of the descriptor, SS_FLAG_CROSSFADE).
    Returns (audiogoal [N,2,out_len], spectrogram [N,65,T4,2] or None)."""
    L = lib()
    rir_bank = np.ascontiguousarray(rir_bank, dtype=np.float32)
    R, _, cap = rir_bank.shape
    nbh_max = max(1, P.ceil_div(cap, P.KB))
    if bucket2 is not None:                              # length-bucketed bank: bank indices >= R live in a second allocation
        bucket2 = np.ascontiguousarray(bucket2, dtype=np.float32)          # [R2, 2, cap2]; rir_len covers both buckets
        assert not spectral and not interleaved and len(rir_len) == R + bucket2.shape[0]
        nbh_max = max(nbh_max, P.ceil_div(bucket2.shape[2], P.KB))
    nby = max(1, P.ceil_div(n_valid, P.KB))
    offs = np.cumsum([0] + [len(s) for s in sources])
    flat = np.concatenate([np.asarray(s, np.float32) for s in sources]).astype(np.float32)
    cache, rows = {}, []

    def slot_of(sound, t0, wrap):
        key = (sound, t0, wrap)
        if key not in cache:
            ws = P.plan_window_set(len(sources[sound]), t0, nbh_max, nby, wrap)
            cache[key] = (sum(len(r) for r in rows), ws)
            rows.append(P.window_desc_rows(ws, int(offs[sound]), len(sources[sound]), wrap))
        return cache[key]

    desc = np.zeros((len(units), 8), np.int32)
    for n, u in enumerate(units):
        if u.get("rir", -1) < 0:
            desc[n] = P.unit_desc_row()
            continue
        s0, ws = slot_of(u["sound"], u["t0"], u.get("wrap", False))
        if crossfade and u.get("last_rir", -1) >= 0:
            d0, dws = slot_of(u["sound"], u["t0"], u.get("last_wrap", u.get("wrap", False)))
            desc[n] = P.unit_desc_row(u["rir"], s0, ws, u["last_rir"], d0, dws)
        elif u.get("dis_rir", -1) >= 0:
            d0, dws = slot_of(u["dis_sound"], u.get("dis_t0", 0), False)
            desc[n] = P.unit_desc_row(u["rir"], s0, ws, u["dis_rir"], d0, dws)
        else:
            desc[n] = P.unit_desc_row(u["rir"], s0, ws)
    wd = np.concatenate(rows) if rows else np.zeros((0, 4), np.int32)
    wd = np.ascontiguousarray(wd, np.int32)
    spec = np.zeros((max(1, len(wd)), P.SPEC_FLOATS), np.float32)
    rc = (L.hs_source_windows32 if core32 else L.hs_source_windows)(_p(flat, ctypes.c_float), _p(wd, ctypes.c_int),
                                                                    _p(spec, ctypes.c_float), len(wd))
    assert rc == 0, rc
    N = len(units)
    out = np.full((N, 2, out_len), np.nan, np.float32)
    t4 = P.spectrogram_shape(out_len)[1]
    sg = np.full((N, 65, t4, 2), np.nan, np.float32)
    rl = np.ascontiguousarray(rir_len, np.int32)
    if interleaved:
        bank = np.ascontiguousarray(rir_bank.transpose(0, 2, 1))     # [R, cap, 2] wav layout
        us, cs, es = 2 * cap, 1, 2
    else:
        bank, us, cs, es = rir_bank, 2 * cap, cap, 1
    no_dis = not any(u.get("dis_rir", -1) >= 0 for u in units)
    simple = int(simple and no_dis and cap <= P.KB and nby == 1)
    if bucket2 is not None:
        in_b0 = all(max(u.get("rir", -1), u.get("dis_rir", -1), u.get("last_rir", -1)) < R for u in units)
        simple = int(simple and in_b0)                   # SS_FLAG_FIRST_BUCKET: the loop-free kernel only sees bucket 0
        L.hs_set_bucket2(_p(bucket2, ctypes.c_float), R, int(bucket2.shape[2]))
    if crossfade:
        simple = 2
    if core32:                                           # 512-thread core: the loop-free row kernel only
        assert simple == 1 and not spectral and not row_wgs and not persist
        rc = L.hs_conv32(int(fuse), _p(spec, ctypes.c_float), _p(bank, ctypes.c_float), _p(rl, ctypes.c_int),
                         _p(desc, ctypes.c_int), _p(out, ctypes.c_float) if want_audiogoal else None,
                         _p(sg, ctypes.c_float) if fuse else None, N, ctypes.c_longlong(us), cs, es, cap, n_valid, out_len,
                         pad_mode, int(tab))
        assert rc == 0, rc
        if want_spectrogram and not fuse:
            L.hs_set_spec_n_valid(int(n_valid))
            rc = L.hs_spectrogram(_p(out, ctypes.c_float), _p(sg, ctypes.c_float), N, out_len, pad_mode, 1)
            assert rc == 0, rc
        return (out if want_audiogoal else None), (sg if (fuse or want_spectrogram) else None)
    if parts_log2:                                       # fused rows rendered by 2^k workgroups each (ConvParams::parts_log2)
        assert (fuse or row_wgs) and not core32 and not persist
        L.hs_set_parts_log2(int(parts_log2))
    if row_wgs:                                          # k_obs_rows: fused rows of 2-3 blocks, `row_wgs` persistent workgroups
        assert out_len > P.KB and not (crossfade and spectral)
        hb, hspec = 0, None
        if spectral:
            assert not interleaved
            hb = P.ceil_div(cap, P.KB)
            hspec = np.zeros((R, 2, hb, P.SPEC_FLOATS), np.float32)
            rc = L.hs_rir_spectra(_p(rir_bank, ctypes.c_float), _p(hspec, ctypes.c_float), R, ctypes.c_longlong(2 * cap), cap, cap)
            assert rc == 0, rc
        if row_blocks:                                   # k_obs_blocks: one workgroup per output block of a row
            L.hs_set_obs_blocks(int(row_blocks))         # (2: the workgroups run in REVERSE order - no hand-off ever arrives)
        rc = L.hs_obs_rows(_p(spec, ctypes.c_float), _p(bank, ctypes.c_float), _p(hspec, ctypes.c_float) if spectral else None,
                           _p(rl, ctypes.c_int), _p(desc, ctypes.c_int), _p(out, ctypes.c_float) if want_audiogoal else None,
                           _p(sg, ctypes.c_float), int(N), ctypes.c_longlong(us), int(cs), int(es), int(cap), int(hb), int(n_valid),
                           int(out_len), int(pad_mode), int(row_wgs), int(no_dis and not crossfade), int(row_stash), int(crossfade))
        assert rc == 0, rc
        return (out if want_audiogoal else None), sg
    if spectral:                                         # spectral RIR bank (ss_rir_spectra_f32 + k_conv_spec)
        assert not crossfade and not interleaved
        hb = P.ceil_div(cap, P.KB)
        hspec = np.zeros((R, 2, hb, P.SPEC_FLOATS), np.float32)
        rc = L.hs_rir_spectra(_p(rir_bank, ctypes.c_float), _p(hspec, ctypes.c_float), R, ctypes.c_longlong(2 * cap), cap, cap)
        assert rc == 0, rc
        rc = L.hs_conv_spec(int(fuse), simple, _p(spec, ctypes.c_float), _p(hspec, ctypes.c_float), _p(rl, ctypes.c_int),
                            _p(desc, ctypes.c_int), _p(out, ctypes.c_float), _p(sg, ctypes.c_float) if fuse else None,
                            N, hb, n_valid, out_len, pad_mode, persist)
        assert rc == 0, rc
        if want_spectrogram and not fuse:
            L.hs_set_spec_n_valid(int(n_valid))             # (as the library's two-launch path: rows known zero from n_valid on)
            rc = L.hs_spectrogram(_p(out, ctypes.c_float), _p(sg, ctypes.c_float), N, out_len, pad_mode, 1)
            assert rc == 0, rc
        return out, (sg if (fuse or want_spectrogram) else None)
    rc = L.hs_conv(int(fuse), simple, _p(spec, ctypes.c_float), _p(bank, ctypes.c_float), _p(rl, ctypes.c_int),
                   _p(desc, ctypes.c_int), _p(out, ctypes.c_float), _p(sg, ctypes.c_float) if fuse else None,
                   N, ctypes.c_longlong(us), cs, es, cap, n_valid, out_len, pad_mode, persist)
    assert rc == 0, rc
    if want_spectrogram and not fuse:
        L.hs_set_spec_n_valid(int(n_valid))
        rc = L.hs_spectrogram(_p(out, ctypes.c_float), _p(sg, ctypes.c_float), N, out_len, pad_mode, 1)
        assert rc == 0, rc
    return out, (sg if (fuse or want_spectrogram) else None)


def spectrogram(x, pad_mode=0, gpw=1):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    N, _, n = x.shape
    sg = np.full((N, 65, P.spectrogram_shape(n)[1], 2), np.nan, np.float32)
    rc = L.hs_spectrogram(_p(x, ctypes.c_float), _p(sg, ctypes.c_float), N, n, pad_mode, gpw)
    assert rc == 0, rc
    return sg


def intensity(x, num_frame=150):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    out = np.full((x.shape[0],), np.nan, np.float32)
    rc = L.hs_intensity(_p(x, ctypes.c_float), _p(out, ctypes.c_float), x.shape[0], x.shape[2], num_frame)
    assert rc == 0, rc
    return out


def logmel(x, sr, n_mels=64, eps=1e-6, pad_mode=0, gpw=1):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    N, _, n = x.shape
    start, w, max_len = P.mel_filterbank_sparse(sr, n_mels)
    start = np.ascontiguousarray(start, np.int32); w = np.ascontiguousarray(w, np.float32)
    out = np.full((N,) + P.logmel_shape(n, n_mels), np.nan, np.float32)
    rc = L.hs_logmel(_p(x, ctypes.c_float), _p(out, ctypes.c_float), N, n, pad_mode, _p(start, ctypes.c_int),
                     _p(w, ctypes.c_float), n_mels, max_len, ctypes.c_float(eps), gpw)
    assert rc == 0, rc
    return out


def gccphat(x, max_lag=32, eps=1e-8, pad_mode=0, gpw=1):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    N, _, n = x.shape
    out = np.full((N, 2 * max_lag + 1, 1 + n // 160), np.nan, np.float32)
    rc = L.hs_gccphat(_p(x, ctypes.c_float), _p(out, ctypes.c_float), N, n, pad_mode, max_lag, ctypes.c_float(eps), gpw)
    assert rc == 0, rc
    return out


def features(x, sr, want=("spectrogram", "logmel", "gccphat"), n_mels=64, mel_eps=1e-6, max_lag=32, gcc_eps=1e-8, pad_mode=0, gpw=1):
    """k_features: every STFT-derived feature from ONE pass over the waveform -> dict of the wanted outputs."""
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    N, _, n = x.shape
    start, w, max_len = P.mel_filterbank_sparse(sr, n_mels)
    start = np.ascontiguousarray(start, np.int32); w = np.ascontiguousarray(w, np.float32)
    T = 1 + n // 160
    out = {}
    if "spectrogram" in want:
        out["spectrogram"] = np.full((N, 65, P.spectrogram_shape(n)[1], 2), np.nan, np.float32)
    if "logmel" in want:
        out["logmel"] = np.full((N,) + P.logmel_shape(n, n_mels), np.nan, np.float32)
    if "gccphat" in want:
        out["gccphat"] = np.full((N, 2 * max_lag + 1, T), np.nan, np.float32)
    ptr = lambda k: _p(out[k], ctypes.c_float) if k in out else None
    rc = L.hs_features(_p(x, ctypes.c_float), N, n, pad_mode, ptr("spectrogram"), ptr("logmel"), _p(start, ctypes.c_int),
                       _p(w, ctypes.c_float), n_mels, max_len, ctypes.c_float(mel_eps), ptr("gccphat"), max_lag,
                       ctypes.c_float(gcc_eps), gpw)
    assert rc == 0, rc
    return out

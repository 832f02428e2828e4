"""Host-side planning for the partitioned FFT convolution (pure Python/numpy, no GPU).

Every windowing branch of the reference collapses to one formula,

    out[c, t] = sum_k rir[c, k] * x[t0 + t - k],      x[n] = 0 for n < 0,

with a branch-specific ``t0`` (``window_start_*`` below).  The kernels evaluate it as a
uniformly-partitioned overlap-save with block ``KB`` = 16384:

    Y_j = sum_i H_i * S_{j-i},   S_m = rFFT_{2KB}( x[t0 + (m-1)KB : t0 + (m+1)KB] )

so a (sound, t0) pair needs the spectra S_m for m in [-(nbh-1), nby-1] that are not
identically zero.  This module computes those window sets and packs the int32 descriptors
``ss_source_windows_f32`` / ``ss_fftconv_binaural_f32`` take (include/ss_hip.h).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

KB = 16384                 # ss_block_len()
SPEC_FLOATS = 2 * 16384    # ss_spec_floats()


def ceil_div(a: int, b: int) -> int:
    return -(-a // b)


# ---- t0 of each reference branch ---------------------------------------------------------------

def window_start_sim(source_len: int, sr: int, audio_index: int) -> int:
    """SoundSpacesSim._compute_audiogoal (soundspaces/simulator.py:629-647).
    1-s clip: full conv, first ``sr`` samples -> t0 = 0.  Multi-second clip: both the 'early'
    (:636-640) and the 'steady' (:641-647) branch equal t0 = index * sr."""
    if source_len == sr:
        return 0
    return audio_index * sr


def window_start_savi_dataset(rir_len: int, sr: int, index: int) -> int:
    """AudioGoalDataset.compute_audiogoal (ss_baselines/savi/pretraining/audiogoal_dataset.py:127-138):
    the steady branch starts one sample earlier than the simulator's and drops the last sample."""
    if index * sr - rir_len < 0:
        return index * sr
    return index * sr - 1


def window_start_continuous(sample_index: int) -> int:
    """ContinuousSoundSpacesSim._convolve_with_rir (soundspaces/continuous_simulator.py:428-456):
    t0 = _current_sample_index for both branches; the steady branch wraps around the clip end."""
    return sample_index


def next_audio_index(audio_index: int, source_len: int, sr: int) -> int:
    """simulator.py:634-635 — advanced only for multi-second sounds."""
    if source_len == sr:
        return audio_index
    return (audio_index + 1) % (source_len // sr)


# ---- window sets ----------------------------------------------------------------------------------

@dataclass(frozen=True)
class WindowSet:
    m_min: int          # first stored partition offset
    count: int          # number of stored offsets (0: the convolution is identically zero)
    starts: tuple       # window start sample for each stored offset


def plan_window_set(source_len: int, t0: int, nbh_max: int, nby: int, wrap: bool = False) -> WindowSet:
    """Offsets m in [-(nbh_max-1), nby-1] whose window x[t0+(m-1)KB : t0+(m+1)KB] is not all zero."""
    lo, hi = -(nbh_max - 1), nby - 1
    limit = 2 * source_len if wrap else source_len        # wrap: indices >= len continue once from 0
    ms = [m for m in range(lo, hi + 1)
          if t0 + (m + 1) * KB > 0 and t0 + (m - 1) * KB < limit]
    if not ms:
        return WindowSet(0, 0, ())
    # contiguous by construction (both conditions are monotone in m)
    return WindowSet(ms[0], len(ms), tuple(t0 + (m - 1) * KB for m in ms))


def window_desc_rows(ws: WindowSet, src_offset: int, source_len: int, wrap: bool = False) -> np.ndarray:
    """int32 [count, 4] rows {src_offset, src_len, start, wrap} for ss_source_windows_f32."""
    rows = np.zeros((ws.count, 4), dtype=np.int32)
    for k, start in enumerate(ws.starts):
        rows[k] = (src_offset, source_len, start, int(wrap))
    return rows


def unit_desc_row(rir_index: int = -1, slot0: int = 0, ws: WindowSet | None = None,
                  dis_rir_index: int = -1, dis_slot0: int = 0, dis_ws: WindowSet | None = None) -> np.ndarray:
    """int32 [8] unit descriptor: two terms {rir index | -1, first slot, m_min, count}."""
    row = np.zeros(8, dtype=np.int32)
    row[0] = -1
    row[4] = -1
    if rir_index >= 0 and ws is not None and ws.count > 0:
        row[0:4] = (rir_index, slot0, ws.m_min, ws.count)
    if dis_rir_index >= 0 and dis_ws is not None and dis_ws.count > 0:
        row[4:8] = (dis_rir_index, dis_slot0, dis_ws.m_min, dis_ws.count)
    return row


def spectrogram_shape(length: int) -> tuple:
    """Observation shape of SpectrogramSensor (soundspaces/tasks/nav.py:76-84): (65, 26, 2) @16 kHz,
    (65, 69, 2) @44.1 kHz."""
    n_frames = 1 + length // 160
    return (65, ceil_div(n_frames, 4), 2)


# ---- log-mel extension (not in the reference; ss_logmel_f32) -------------------------------------------------------
def live_pooled_blocks(n_valid: int, length: int) -> int:
    """Pooled spectrogram columns that can be non-zero when a row of `length` samples is zero from n_valid on
    (csrc/ss_kernels.hpp::live_blocks: column b is zero once 640 b - 256 >= n_valid, provided the right centre padding
    mirrors zeros too)."""
    t4 = spectrogram_shape(length)[1]
    if n_valid > length - 512:
        return t4
    return min(t4, ceil_div(n_valid + 256, 640))


def wide_one_block(out_len: int, n_valid: int, spectral: bool = False) -> bool:
    """Rows longer than one block of which only block 0 is rendered and whose live columns fit the fused phase (SS2.0 steps
    at 44.1 kHz): the library serves them with the fused loop kernel in ONE launch, waveform buffer or not
    (csrc/ss_hip.hip::wide_one_block_ok)."""
    return (not spectral) and out_len > KB and 0 <= n_valid <= KB and live_pooled_blocks(n_valid, out_len) <= 26


def _slaney_mel(f):
    """Hz -> mel on the Slaney / auditory-toolbox scale (linear to 1 kHz, log above): librosa's default."""
    f = np.asarray(f, dtype=np.float64)
    lin = 3.0 * f / 200.0
    return np.where(f >= 1000.0, 15.0 + 27.0 * np.log(np.maximum(f, 1e-30) / 1000.0) / np.log(6.4), lin)


def _slaney_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * np.log(6.4) / 27.0), 200.0 * m / 3.0)


def mel_filterbank_sparse(sr: int, n_mels: int = 64, n_fft: int = 512, fmin: float = 0.0, fmax=None):
    """Area-normalised triangular mel filters in the band-sparse form ss_logmel_f32 takes:
    (start int32 [n_mels], weights float32 [n_mels, max_len], max_len); band j weighs bins start[j] + i.
    Triangles are evaluated per bin from the band edges (each filter is non-zero on one contiguous run of bins)."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    n_bins = 1 + n_fft // 2
    edges = _slaney_hz(np.linspace(_slaney_mel(fmin), _slaney_mel(fmax), n_mels + 2))
    bin_hz = np.arange(n_bins, dtype=np.float64) * (sr / float(n_fft))
    rows, starts = [], []
    for j in range(n_mels):
        lo, ce, hi = edges[j], edges[j + 1], edges[j + 2]
        rise = (bin_hz - lo) / (ce - lo)
        fall = (hi - bin_hz) / (hi - ce)
        tri = np.clip(np.minimum(rise, fall), 0.0, None) * (2.0 / (hi - lo))
        nz = np.nonzero(tri)[0]
        if nz.size == 0:                    # band narrower than one bin (librosa warns: "empty filters")
            starts.append(0); rows.append(np.zeros(4))
        else:
            s0 = int(nz[0]) & ~3            # ABI: starts are multiples of 4 (16-byte LDS reads), leading weights zero
            starts.append(s0); rows.append(tri[s0:nz[-1] + 1])
    max_len = (max(len(r) for r in rows) + 3) & ~3
    w = np.zeros((n_mels, max_len), np.float32)
    for j, r in enumerate(rows):
        w[j, :len(r)] = r
    return np.asarray(starts, np.int32), w, max_len


def logmel_shape(n_samples: int, n_mels: int):
    return (n_mels, 1 + n_samples // 160, 2)

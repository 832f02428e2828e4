"""CPU oracle for the SoundSpaces audio-observation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the reported baseline.

It restates, in numpy + scipy (the libraries the reference itself calls), the
arithmetic of these reference functions (paths relative to the reference
checkout):

* ``soundspaces/simulator.py:608-666``   ``SoundSpacesSim._compute_audiogoal``
* ``soundspaces/simulator.py:678-701``   the two memo caches
* ``soundspaces/tasks/nav.py:86-100``    ``SpectrogramSensor.compute_spectrogram``
* ``soundspaces/continuous_simulator.py:47-53, 413-456``  cross-fade,
  ``_compute_audiogoal`` and ``_convolve_with_rir`` of the SS2.0 simulator
* ``ss_baselines/savi/pretraining/audiogoal_dataset.py:114-156``
* ``ss_baselines/av_wan/avwan_sensors.py:91-100``  ``Intensity``

and, clearly separated at the end of the file, two EXTENSION definitions that have no
counterpart in the reference (``compute_logmel``, ``compute_gcc_phat``): "parity unpinned".

Pinning status
--------------
* Convolution half: PINNED.  ``tests/golden/make_golden.py`` executes the
  reference's own ``_compute_audiogoal`` / ``_convolve_with_rir`` /
  ``crossfade`` source (extracted from the reference checkout at generation
  time, never copied into this repo) on real scipy and stores the outputs;
  ``tests/test_oracle.py`` checks this restatement against those vectors.
* Spectrogram half: the reference calls ``librosa.stft`` and
  ``skimage.measure.block_reduce``; neither library is installable here, so
  ``stft`` / ``block_reduce`` below restate their documented defaults and are
  cross-checked against ``torch.stft`` and ``scipy.signal.ShortTimeFFT``.
  The reference has no golden vectors of its own for this path (it ships no
  tests): "parity unpinned" applies to the librosa/skimage semantics only
  (``pad_mode`` default changed in librosa 0.10: both are implemented).
"""
from __future__ import annotations

import numpy as np
from scipy.signal import fftconvolve, get_window

# --------------------------------------------------------------------------
# A2: SoundSpacesSim._compute_audiogoal  (soundspaces/simulator.py:608-666)
# --------------------------------------------------------------------------

N_FFT = 512          # nav.py:88
HOP_LENGTH = 160     # nav.py:89
WIN_LENGTH = 400     # nav.py:90
POOL = 4             # nav.py:93


def _convolve_channels(source, rir, mode="full"):
    """[fftconvolve(source, rir[:, c]) for c in channels] (simulator.py:630-631)."""
    return np.array([fftconvolve(source, rir[:, c], mode=mode)
                     for c in range(rir.shape[-1])])


def zero_rir(sr):
    """Fallback for an unreadable / empty RIR wav (simulator.py:619-624)."""
    return np.zeros((sr, 2)).astype(np.float32)


def compute_audiogoal(source, rir, sr, audio_index=0, silent=False,
                      distractor=None, distractor_rir=None):
    """Restates simulator.py:608-666.

    source        float32 [S] mono clip already resampled to ``sr``
    rir           float32 [L, 2] (wav layout) — pass ``zero_rir(sr)`` for the
                  unreadable-file fallback
    audio_index   the value of ``self._audio_index`` *before* the call; the
                  caller advances it ``(index + 1) % (S // sr)`` (line 635)
    silent        ``_episode_step_count > _duration`` (line 610)
    distractor    optional float32 [Sd] clip, distractor_rir float32 [Ld, 2]
    returns       [2, sr] (float64 zeros if silent, float32 otherwise)
    """
    if silent:
        return np.zeros((2, sr))                                  # :612
    if source.shape[0] == sr:                                     # :629
        conv = _convolve_channels(source, rir)
        audiogoal = conv[:, :sr]                                  # :632
    else:
        index = audio_index                                       # :634
        if index * sr - rir.shape[0] < 0:                         # :636
            seg = source[: (index + 1) * sr]
            conv = _convolve_channels(seg, rir)
            audiogoal = conv[:, index * sr:(index + 1) * sr]      # :640
        else:
            seg = source[index * sr - rir.shape[0] + 1:(index + 1) * sr]   # :643
            audiogoal = _convolve_channels(seg, rir, mode="valid")         # :645
    if distractor is not None:                                    # :649
        dconv = _convolve_channels(distractor, distractor_rir)
        audiogoal = audiogoal + dconv[:, :sr]                     # :664
    return audiogoal


def next_audio_index(audio_index, source_len, sr):
    """simulator.py:635 — only advanced for multi-second sounds."""
    if source_len == sr:
        return audio_index
    return (audio_index + 1) % (source_len // sr)


# --------------------------------------------------------------------------
# The same arithmetic as ONE formula (what the HIP kernels implement):
#   out[c, t] = sum_k rir[k, c] * x[t0 + t - k],   x[n] = 0 for n < 0
# All three windowing branches of _compute_audiogoal, the SS2.0 branches and
# the savi dataset variant are this formula with a branch-specific t0.
# --------------------------------------------------------------------------

def window_start(source_len, rir_len, sr, audio_index, variant="sim"):
    """t0 of the unified formula for each reference branch."""
    if source_len == sr and variant == "sim":
        return 0                                  # simulator.py:629-632
    if variant == "sim":
        return audio_index * sr                   # :636-647 (both branches)
    if variant == "savi":
        # audiogoal_dataset.py:127-138: early branch == sim; steady branch
        # starts one sample earlier and drops the last output sample.
        if audio_index * sr - rir_len < 0:
            return audio_index * sr
        return audio_index * sr - 1
    raise ValueError(variant)


def conv_window_direct(source, rir, t0, out_len, wrap=False):
    """Direct O(L*T) evaluation of the unified formula in float64.

    Independent of scipy's FFT path; used to validate both the oracle and
    the window descriptors on small cases.  ``wrap`` = continuous-simulator
    wrap-around of the source (continuous_simulator.py:441-445).
    """
    S = source.shape[0]
    L = rir.shape[0]
    x = source.astype(np.float64)
    out = np.zeros((rir.shape[1], out_len))
    n = t0 + np.arange(out_len)[:, None] - np.arange(L)[None, :]   # [T, L]
    if wrap:
        valid = n >= 0
        idx = np.where(n >= S, n - S, n)
        valid &= idx < S
    else:
        valid = (n >= 0) & (n < S)
        idx = n
    xs = np.where(valid, x[np.clip(idx, 0, S - 1)], 0.0)
    for c in range(rir.shape[1]):
        out[c] = xs @ rir[:, c].astype(np.float64)
    return out


def conv_window_fft(source, rir, t0, out_len):
    """The unified formula evaluated with scipy (fast): full convolution of x[:t0+out_len] sliced at t0.
    Equals every non-wrapping reference branch for the matching t0 (tests/test_oracle.py)."""
    seg = np.asarray(source)[: max(0, t0 + out_len)]
    if rir.shape[0] == 0 or seg.shape[0] == 0:
        return np.zeros((rir.shape[1], out_len), np.float32)
    full = _convolve_channels(seg, rir)
    out = np.zeros((rir.shape[1], out_len), full.dtype)
    avail = full[:, t0:t0 + out_len]
    out[:, :avail.shape[1]] = avail
    return out


# --------------------------------------------------------------------------
# A3: SpectrogramSensor.compute_spectrogram  (soundspaces/tasks/nav.py:86-100)
# --------------------------------------------------------------------------

def stft_window():
    """librosa.stft window: get_window('hann', 400, fftbins=True) centre-padded
    to n_fft=512 (56 zeros each side)."""
    w = get_window("hann", WIN_LENGTH, fftbins=True)
    lpad = (N_FFT - WIN_LENGTH) // 2
    return np.pad(w, (lpad, N_FFT - WIN_LENGTH - lpad))


def stft(signal, pad_mode="reflect"):
    """librosa.stft(signal, n_fft=512, hop_length=160, win_length=400) with the
    library defaults window='hann', center=True.  pad_mode: 'reflect'
    (librosa < 0.10, the reference's era) or 'constant' (librosa >= 0.10).
    Returns complex [257, 1 + len // 160]; complex64 for float32 input."""
    signal = np.asarray(signal)
    out_dtype = np.complex64 if signal.dtype == np.float32 else np.complex128
    y = np.pad(signal, N_FFT // 2, mode=pad_mode)
    n_frames = 1 + (y.shape[0] - N_FFT) // HOP_LENGTH
    idx = HOP_LENGTH * np.arange(n_frames)[:, None] + np.arange(N_FFT)[None, :]
    frames = y[idx] * stft_window()[None, :].astype(y.dtype)
    spec = np.fft.rfft(frames.astype(np.float64), axis=1).T
    return spec.astype(out_dtype)


def block_reduce_mean(a, block=(POOL, POOL)):
    """skimage.measure.block_reduce(a, block, np.mean): pad with cval=0 up to a
    multiple of the block, then mean over each block (pad included)."""
    pr = (-a.shape[0]) % block[0]
    pc = (-a.shape[1]) % block[1]
    a = np.pad(a, ((0, pr), (0, pc)), mode="constant", constant_values=0)
    r, c = a.shape[0] // block[0], a.shape[1] // block[1]
    return a.reshape(r, block[0], c, block[1]).mean(axis=(1, 3))


def compute_spectrogram(audio_data, pad_mode="reflect"):
    """Restates nav.py:86-100.  audio_data [2, T] -> [65, ceil((1+T//160)/4), 2]."""
    def compute_stft(sig):
        mag = np.abs(stft(sig, pad_mode=pad_mode))          # nav.py:92
        return block_reduce_mean(mag)                       # nav.py:93
    c1 = np.log1p(compute_stft(audio_data[0]))              # nav.py:96
    c2 = np.log1p(compute_stft(audio_data[1]))              # nav.py:97
    return np.stack([c1, c2], axis=-1)                      # nav.py:98


def spectrogram_shape(sr):
    """Observation-space shape KAT (nav.py:76-84): (65,26,2) @16k, (65,69,2) @44.1k."""
    t = 1 + sr // HOP_LENGTH
    return ((N_FFT // 2 + 1 + POOL - 1) // POOL, (t + POOL - 1) // POOL, 2)


# --------------------------------------------------------------------------
# A1: the two memo caches  (soundspaces/simulator.py:678-701, reset :395-397)
# --------------------------------------------------------------------------

class CachedSimAudio:
    """Eager-mode semantics of get_current_{audiogoal,spectrogram}_observation:
    caches keyed (source_idx, receiver_idx, azimuth), bypassed with a distractor."""

    def __init__(self, has_distractor=False):
        self.has_distractor = has_distractor
        self._audiogoal_cache = {}
        self._spectrogram_cache = {}

    def clear(self):                                         # simulator.py:395-397
        self._audiogoal_cache = {}
        self._spectrogram_cache = {}

    def audiogoal(self, key, compute):                       # :678-688
        if self.has_distractor:
            return compute()
        if key not in self._audiogoal_cache:
            self._audiogoal_cache[key] = compute()
        return self._audiogoal_cache[key]

    def spectrogram(self, key, compute, audiogoal2spectrogram):   # :690-701
        if self.has_distractor:
            return audiogoal2spectrogram(self.audiogoal(key, compute))
        if key not in self._spectrogram_cache:
            self._spectrogram_cache[key] = audiogoal2spectrogram(self.audiogoal(key, compute))
        return self._spectrogram_cache[key]


# --------------------------------------------------------------------------
# A5: ContinuousSoundSpacesSim  (soundspaces/continuous_simulator.py)
# --------------------------------------------------------------------------

def crossfade(x1, x2, sr):
    """continuous_simulator.py:47-53."""
    n = int(0.05 * sr)
    w2 = np.arange(n + 1) / n
    w1 = np.flip(w2)
    return np.concatenate([x1[:, :n + 1] * w1 + x2[:, :n + 1] * w2, x2[:, n + 1:]], axis=1)


def tile_short_source(source, sr):
    """continuous_simulator.py:408-410: 1-s sounds are tiled x3 at load."""
    if source.shape[0] // sr == 1:
        return np.concatenate([source] * 3, axis=0)
    return source


def convolve_with_rir(source, rir, sr, sample_index, step_time):
    """continuous_simulator.py:428-456.  Returns [2, sr], first int(sr*step_time)
    samples non-zero."""
    num_sample = int(sr * step_time)
    index = sample_index
    if index - rir.shape[0] < 0:                                        # :433
        seg = source[: index + num_sample]
        conv = _convolve_channels(seg, rir)
        audiogoal = conv[:, index: index + num_sample]                  # :437
    else:
        if index + num_sample < source.shape[0]:                        # :440
            seg = source[index - rir.shape[0] + 1: index + num_sample]
        else:
            wrap = index + num_sample - source.shape[0]                 # :443
            seg = np.concatenate([source[index - rir.shape[0] + 1:], source[:wrap]])
        audiogoal = _convolve_channels(seg, rir, mode="valid")          # :447
    return np.pad(audiogoal, [(0, 0), (0, sr - audiogoal.shape[1])])    # :454


def compute_audiogoal_continuous(source, rir, sr, sample_index, step_time,
                                 last_rir=None, use_crossfade=False, silent=False):
    """continuous_simulator.py:413-426."""
    if silent:
        return np.zeros((2, sr))
    audiogoal = convolve_with_rir(source, rir, sr, sample_index, step_time)
    if use_crossfade and last_rir is not None:
        prev = convolve_with_rir(source, last_rir, sr, sample_index, step_time)
        audiogoal = crossfade(prev, audiogoal, sr)
    return audiogoal


def next_sample_index(sample_index, sr, step_time, source_len):
    """continuous_simulator.py:389-390."""
    return int(sample_index + sr * step_time) % source_len


# --------------------------------------------------------------------------
# savi AudioGoalDataset variant  (audiogoal_dataset.py:114-140)
# --------------------------------------------------------------------------

def compute_audiogoal_savi_dataset(source, rir, sr, index):
    if index * sr - rir.shape[0] < 0:                                   # :127
        seg = source[: (index + 1) * sr]
        conv = _convolve_channels(seg, rir)
        return conv[:, index * sr:(index + 1) * sr]
    seg = source[index * sr - rir.shape[0]:(index + 1) * sr]           # :134
    conv = _convolve_channels(seg, rir, mode="valid")
    return conv[:, :-1]                                                 # :138


# --------------------------------------------------------------------------
# A7: av_wan Intensity  (ss_baselines/av_wan/avwan_sensors.py:91-100)
# --------------------------------------------------------------------------

def intensity(audiogoal, num_frame=150):
    nonzero_idx = np.min((audiogoal > 0.1 * audiogoal.max()).argmax(axis=1))
    impulse = audiogoal[:, nonzero_idx: nonzero_idx + num_frame]
    return [np.mean(impulse ** 2)]


# --------------------------------------------------------------------------
# Seeded synthetic inputs shared by tests and bench.
# --------------------------------------------------------------------------

def synth_rir(rng, sr, length=None, n=1):
    """h[c, k] = g * N(0,1) * exp(-6.9 k / (RT60 * sr)) + direct-path impulse,
    per-ear gain / delay offsets <= 0.7 ms, peak-normalised to 0.5.
    Returns float32 [n, 2, L] (planar)."""
    L = sr if length is None else length
    k = np.arange(L)
    out = np.zeros((n, 2, L), dtype=np.float32)
    for i in range(n):
        rt60 = rng.uniform(0.2, 0.8)
        n0 = int(rng.uniform(0, 0.01 * sr))
        for c in range(2):
            g = rng.uniform(0.5, 1.0)
            d = n0 + int(rng.uniform(0, 0.0007 * sr))
            h = g * rng.standard_normal(L) * np.exp(-6.9 * k / (rt60 * sr)) * 0.1
            h[:d] = 0.0
            if d < L:
                h[d] += g
            out[i, c] = h
        out[i] *= 0.5 / np.abs(out[i]).max()
    return out


def synth_sources(rng, sr, k=4, seconds=1):
    """float32 [k, seconds*sr] white noise in U(-1, 1)."""
    return rng.uniform(-1.0, 1.0, size=(k, seconds * sr)).astype(np.float32)


def relerr(got, ref):
    """max |got - ref| / max |ref|  — the parity figure of SURVEY.md 8(c)."""
    ref = np.asarray(ref)
    got = np.asarray(got)
    wide = np.complex128 if (np.iscomplexobj(ref) or np.iscomplexobj(got)) else np.float64
    ref = ref.astype(wide)
    den = np.abs(ref).max()
    num = np.abs(got.astype(wide) - ref).max()
    return num / den if den > 0 else num


# --------------------------------------------------------------------------
# EXTENSION (BASELINE.json north_star "log-mel"): NOT in the
# reference.  There is no reference implementation to pin against ("parity unpinned");
# this restates the textbook definition with librosa's documented defaults
# (librosa.filters.mel: Slaney mel scale, htk=False, norm='slaney') on top of the same
# STFT framing as compute_spectrogram, and is cross-checked in tests against an
# independent dense construction.
# --------------------------------------------------------------------------

def hz_to_mel(f):
    """Slaney (auditory toolbox) mel scale: linear below 1 kHz, logarithmic above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mel)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_mels=64, n_fft=N_FFT, fmin=0.0, fmax=None):
    """Triangular filters [n_mels, 1 + n_fft//2], area-normalised ('slaney')."""
    fmax = sr / 2.0 if fmax is None else fmax
    fft_f = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, fft_f.size))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


def compute_logmel(audio_data, sr, n_mels=64, eps=1e-6, pad_mode="reflect"):
    """[2, T] -> log(mel(|STFT|^2) + eps), [n_mels, 1 + T//160, 2] channel-last (no pooling)."""
    fb = mel_filterbank(sr, n_mels)
    chans = []
    for c in range(2):
        p = np.abs(stft(audio_data[c], pad_mode=pad_mode).astype(np.complex128)) ** 2
        chans.append(np.log(fb @ p + eps))
    return np.stack(chans, axis=-1)


def compute_gcc_phat(audio_data, max_lag=32, eps=1e-8, pad_mode="reflect"):
    """EXTENSION (not in the reference; see the note above compute_logmel).  Generalised cross-correlation with
    phase transform between the two ears, per STFT frame:
        G[k] = X_l[k] conj(X_r[k]);  g = irfft(G / (|G| + eps), 512);  out[i] = g[(i - max_lag) mod 512]
    [2, T] -> [2*max_lag + 1, 1 + T//160] (lag -max_lag .. +max_lag; positive lag = left ear delayed)."""
    xl = stft(audio_data[0], pad_mode=pad_mode).astype(np.complex128)
    xr = stft(audio_data[1], pad_mode=pad_mode).astype(np.complex128)
    g = xl * np.conj(xr)
    g = g / (np.abs(g) + eps)
    cc = np.fft.irfft(g, n=N_FFT, axis=0)
    return np.concatenate([cc[N_FFT - max_lag:], cc[:max_lag + 1]], axis=0)

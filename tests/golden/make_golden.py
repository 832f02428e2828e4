#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE.

The reference package cannot be imported here (habitat, librosa, skimage, gym
are not installed), but the arithmetic of its hot path lives in a handful of
methods that only need numpy + scipy.  This script extracts those methods'
source text from the reference checkout with ``ast`` at generation time
(nothing is copied into this repository), ``exec``s them against stub ``self``
objects, and stores their outputs:

* ``SoundSpacesSim._compute_audiogoal``            soundspaces/simulator.py:608-666
* ``ContinuousSoundSpacesSim._compute_audiogoal`` / ``_convolve_with_rir`` and
  module-level ``crossfade``                       soundspaces/continuous_simulator.py:47-53,413-456
* ``AudioGoalDataset.compute_audiogoal``           ss_baselines/savi/pretraining/audiogoal_dataset.py:114-140
* ``Intensity.get_observation``                    ss_baselines/av_wan/avwan_sensors.py:91-100
* ``SpectrogramSensor.compute_spectrogram``        soundspaces/tasks/nav.py:86-100
  -- executed with ``librosa.stft`` / ``block_reduce`` bound to the oracle's
  restatements (those two libraries are absent), so this last one pins the
  composition (abs -> 4x4 pool -> log1p -> channel-last stack), not librosa.

Inputs are regenerated from seeds by ``oracle.ss_oracle.synth_*`` (recorded in
the npz) except the one real clip, ``res/singing.wav`` resampled to 16 kHz,
whose first second is stored verbatim.

Run from the repo root in the build container:  python tests/golden/make_golden.py
"""
import ast
import json
import logging
import os
import sys
import textwrap
import types

import numpy as np
from scipy.io import wavfile
from scipy.signal import fftconvolve, resample_poly

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ss_oracle as O  # noqa: E402

REF = os.environ.get("SS_REFERENCE", "/root/reference")


def extract(path, cls, func):
    """Source text of ``cls.func`` (or module-level ``func`` if cls is None)."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == func)
    seg = ast.get_source_segment(src, node)
    lines = src.splitlines()[node.lineno - 1: node.end_lineno]
    text = textwrap.dedent("\n".join(lines))
    # drop decorators such as @staticmethod (we call the plain function)
    text = "\n".join(l for l in text.splitlines() if not l.startswith("@"))
    assert seg is not None
    return text


class NS(types.SimpleNamespace):
    pass


class FakeWav:
    """stands in for scipy.io.wavfile: the RIR 'files' are in-memory arrays."""
    def __init__(self):
        self.files = {}

    def read(self, path):
        v = self.files[path]
        if isinstance(v, Exception):
            raise v
        return 16000, v


def load_fn(path, cls, func, extra):
    ns = {"np": np, "fftconvolve": fftconvolve, "os": os, "logging": logging}
    ns.update(extra)
    exec(extract(path, cls, func), ns)
    return ns[func]


def main():
    out = {}
    meta = {}
    fw = FakeWav()
    sim_audiogoal = load_fn("soundspaces/simulator.py", "SoundSpacesSim", "_compute_audiogoal",
                            {"wavfile": fw})

    def run_sim(sr, source, rir, audio_index=0, step=0, duration=500, distractor=None,
                distractor_rir=None, rir_error=None):
        fw.files.clear()
        s = NS()
        s.config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, HAS_DISTRACTOR_SOUND=distractor is not None),
                      USE_RENDERED_OBSERVATIONS=True)
        s._episode_step_count = step
        s._duration = duration
        s.binaural_rir_dir = "rirs"
        s.azimuth_angle = 90
        s._receiver_position_index = 3
        s._source_position_index = 7
        s._distractor_position_index = 11
        s.current_source_sound = source
        s._audio_index = audio_index
        s._audio_length = source.shape[0] // sr
        s._current_distractor_sound = "d.wav"
        s._source_sound_dict = {"d.wav": distractor}
        fw.files[os.path.join("rirs", "90", "3_7.wav")] = rir_error if rir_error is not None else rir
        fw.files[os.path.join("rirs", "90", "3_11.wav")] = distractor_rir
        y = sim_audiogoal(s)
        return y, s._audio_index

    rng = np.random.default_rng(0)
    sr = 16000
    wav_layout = lambda h: np.ascontiguousarray(h.T)          # planar [2,L] -> wav [L,2]

    # ---- one real clip: res/singing.wav (48 kHz int16 mono) -> 16 kHz, first second
    fs, sing = wavfile.read(os.path.join(REF, "res", "singing.wav"))
    sing = sing.astype(np.float64) / 32768.0
    sing16 = resample_poly(sing, sr, fs)[:sr].astype(np.float32)
    out["singing_16k"] = sing16

    cases = []

    def add(name, audiogoal, **params):
        out[name + "/audiogoal"] = np.asarray(audiogoal)
        cases.append(name)
        meta[name] = params

    # ---- A2 (iii) 1-s clip, L = sr
    src = O.synth_sources(np.random.default_rng(1), sr, k=3, seconds=1)
    rir = O.synth_rir(np.random.default_rng(2), sr, n=3)
    y, _ = run_sim(sr, src[0], wav_layout(rir[0]))
    add("clip1s", y, sr=sr, src_seed=1, src_k=3, src_sel=0, rir_seed=2, rir_n=3, rir_sel=0, rir_len=sr)
    # ---- ragged L
    Lr = 5921
    rir_r = O.synth_rir(np.random.default_rng(3), sr, length=Lr, n=1)
    y, _ = run_sim(sr, src[1], wav_layout(rir_r[0]))
    add("clip1s_ragged", y, sr=sr, src_seed=1, src_k=3, src_sel=1, rir_seed=3, rir_n=1, rir_sel=0, rir_len=Lr)
    # ---- real clip
    y, _ = run_sim(sr, sing16, wav_layout(rir[1]))
    add("clip1s_singing", y, sr=sr, src="singing_16k", rir_seed=2, rir_n=3, rir_sel=1, rir_len=sr)
    # ---- unreadable RIR (ValueError) -> zero RIR -> zeros
    y, _ = run_sim(sr, src[0], None, rir_error=ValueError("bad wav"))
    add("zero_rir", y, sr=sr)
    # ---- empty RIR file
    y, _ = run_sim(sr, src[0], np.zeros((0, 2), np.float32))
    add("empty_rir", y, sr=sr)
    # ---- silent
    y, _ = run_sim(sr, src[0], wav_layout(rir[0]), step=501, duration=500)
    add("silent", y, sr=sr)
    # ---- multi-second (iv)/(v): 5-s source; L = sr and L = 1.5 sr
    src5 = O.synth_sources(np.random.default_rng(4), sr, k=1, seconds=5)[0]
    rir15 = O.synth_rir(np.random.default_rng(5), sr, length=sr + sr // 2, n=1)
    for idx in range(5):
        y, nxt = run_sim(sr, src5, wav_layout(rir[2]), audio_index=idx)
        add(f"multi_L1.0_i{idx}", y, sr=sr, src_seed=4, seconds=5, rir_seed=2, rir_n=3, rir_sel=2,
            rir_len=sr, audio_index=idx, next_index=int(nxt))
    for idx in (0, 1, 2, 4):
        y, nxt = run_sim(sr, src5, wav_layout(rir15[0]), audio_index=idx)
        add(f"multi_L1.5_i{idx}", y, sr=sr, src_seed=4, seconds=5, rir_seed=5, rir_n=1, rir_sel=0,
            rir_len=sr + sr // 2, audio_index=idx, next_index=int(nxt))
    # ---- distractor (vi): 1-s source + 1-s distractor through a second RIR
    y, _ = run_sim(sr, src[0], wav_layout(rir[0]), distractor=src[2], distractor_rir=wav_layout(rir[1]))
    add("distractor", y, sr=sr, src_seed=1, src_k=3, src_sel=0, dis_sel=2, rir_seed=2, rir_n=3, rir_sel=0,
        dis_rir_sel=1)
    # ---- 44.1 kHz
    sr2 = 44100
    src44 = O.synth_sources(np.random.default_rng(6), sr2, k=1, seconds=1)[0]
    rir44 = O.synth_rir(np.random.default_rng(7), sr2, n=1)
    y, _ = run_sim(sr2, src44, wav_layout(rir44[0]))
    add("clip1s_44k", y, sr=sr2, src_seed=6, rir_seed=7)

    # ---- 44.1 kHz beyond the 1-s clip (round 3: rows of three partition blocks go through the fused k_obs_rows): a 3-s
    # source in the early and the steady branch, a 1.5-s RIR (5 blocks), a ragged RIR, a distractor
    src44_3 = O.synth_sources(np.random.default_rng(10), sr2, k=1, seconds=3)[0]
    rir44b = O.synth_rir(np.random.default_rng(11), sr2, n=2)
    for idx in (0, 1, 2):
        y, nxt = run_sim(sr2, src44_3, wav_layout(rir44b[0]), audio_index=idx)
        add(f"multi_L1.0_i{idx}_44k", y, sr=sr2, src_seed=10, seconds=3, rir_seed=11, rir_n=2, rir_sel=0, rir_len=sr2,
            audio_index=idx, next_index=int(nxt))
    rir44_15 = O.synth_rir(np.random.default_rng(12), sr2, length=66150, n=1)
    y, nxt = run_sim(sr2, src44_3, wav_layout(rir44_15[0]), audio_index=2)
    add("multi_L1.5_i2_44k", y, sr=sr2, src_seed=10, seconds=3, rir_seed=12, rir_n=1, rir_sel=0, rir_len=66150,
        audio_index=2, next_index=int(nxt))
    rir44_r = O.synth_rir(np.random.default_rng(13), sr2, length=30011, n=1)
    srcs44 = O.synth_sources(np.random.default_rng(14), sr2, k=2, seconds=1)
    y, _ = run_sim(sr2, srcs44[0], wav_layout(rir44_r[0]))
    add("clip1s_ragged_44k", y, sr=sr2, src_seed=14, src_k=2, src_sel=0, rir_seed=13, rir_n=1, rir_sel=0, rir_len=30011)
    y, _ = run_sim(sr2, srcs44[0], wav_layout(rir44b[0]), distractor=srcs44[1], distractor_rir=wav_layout(rir44b[1]))
    add("distractor_44k", y, sr=sr2, src_seed=14, src_k=2, src_sel=0, dis_sel=1, rir_seed=11, rir_n=2, rir_sel=0,
        dis_rir_sel=1)

    # ---- A5 continuous simulator
    crossfade = load_fn("soundspaces/continuous_simulator.py", None, "crossfade", {})
    cws = load_fn("soundspaces/continuous_simulator.py", "ContinuousSoundSpacesSim", "_convolve_with_rir", {})
    cag = load_fn("soundspaces/continuous_simulator.py", "ContinuousSoundSpacesSim", "_compute_audiogoal",
                  {"crossfade": crossfade})

    def run_cont(sr, source, rir, sample_index, last_rir=None, use_crossfade=False, step_time=0.25):
        s = NS()
        s.config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, CROSSFADE=use_crossfade), STEP_TIME=step_time)
        s._episode_step_count = 0
        s._duration = 500
        s._prev_sim_obs = {"audio_sensor": rir.T.tolist()}       # habitat_sim hands [2][L] lists
        s._last_rir = last_rir
        s._current_sample_index = sample_index
        s.current_source_sound = source
        s._convolve_with_rir = lambda r: cws(s, r)
        return cag(s)

    src3 = O.tile_short_source(src[0], sr)                       # 1-s clip tiled x3
    rc = O.synth_rir(np.random.default_rng(8), sr, length=9000, n=2)
    for name, si in (("early", 1234), ("steady", 20000), ("wrap", 3 * sr - 1500)):
        y = run_cont(sr, src3, wav_layout(rc[0]), si)
        add(f"cont_{name}", y, sr=sr, src_seed=1, src_k=3, src_sel=0, rir_seed=8, rir_len=9000, rir_n=2,
            rir_sel=0, sample_index=si, step_time=0.25)
    # _last_rir is np.transpose(np.array(<lists>)) in the simulator (:384) -> float64
    y = run_cont(sr, src3, wav_layout(rc[0]), 20000, last_rir=wav_layout(rc[1]).astype(np.float64),
                 use_crossfade=True)
    add("cont_crossfade", y, sr=sr, src_seed=1, src_k=3, src_sel=0, rir_seed=8, rir_len=9000, rir_n=2,
        rir_sel=0, last_rir_sel=1, sample_index=20000, step_time=0.25)

    # ---- SS2.0 at 44.1 kHz: 0.25-s steps = 11025 samples of a 44100-sample row (one convolved block, two zero blocks)
    src3_44 = O.tile_short_source(srcs44[0], sr2)
    rc44 = O.synth_rir(np.random.default_rng(15), sr2, length=20000, n=1)
    for name, si in (("early", 3000), ("steady", 50000)):
        y = run_cont(sr2, src3_44, wav_layout(rc44[0]), si)
        add(f"cont_{name}_44k", y, sr=sr2, src_seed=14, src_k=2, src_sel=0, rir_seed=15, rir_len=20000, rir_n=1,
            rir_sel=0, sample_index=si, step_time=0.25)

    # ---- ... and cross-faded (round 3, late: k_obs_rows renders block 0 twice and blends on the CU): previous RIR = the
    # bank's second entry, ramp of int(0.05 * 44100) + 1 = 2206 samples
    rc44x = O.synth_rir(np.random.default_rng(16), sr2, length=20000, n=2)
    y = run_cont(sr2, src3_44, wav_layout(rc44x[0]), 50000, last_rir=wav_layout(rc44x[1]).astype(np.float64),
                 use_crossfade=True)
    add("cont_crossfade_44k", y, sr=sr2, src_seed=14, src_k=2, src_sel=0, rir_seed=16, rir_len=20000, rir_n=2, rir_sel=0,
        last_rir_sel=1, sample_index=50000, step_time=0.25)

    # ---- 48 kHz (round 4: the one-launch kernel for rows with one rendered block serves 44.1 AND 48 kHz steps; the cross-fade
    # ramp of int(0.05 * 48000) + 1 = 2401 samples is the longest the kernels hold): a plain step in either branch, a
    # cross-faded step, and a SoundSpaces-1.0 multi-second clip (three blocks per row)
    sr3 = 48000
    srcs48 = O.synth_sources(np.random.default_rng(24), sr3, k=2, seconds=1)
    src3_48 = O.tile_short_source(srcs48[0], sr3)
    rc48 = O.synth_rir(np.random.default_rng(25), sr3, length=26000, n=2)
    for name, si in (("early", 5000), ("steady", 61000)):
        y = run_cont(sr3, src3_48, wav_layout(rc48[0]), si)
        add(f"cont_{name}_48k", y, sr=sr3, src_seed=24, src_k=2, src_sel=0, rir_seed=25, rir_len=26000, rir_n=2,
            rir_sel=0, sample_index=si, step_time=0.25)
    y = run_cont(sr3, src3_48, wav_layout(rc48[0]), 61000, last_rir=wav_layout(rc48[1]).astype(np.float64), use_crossfade=True)
    add("cont_crossfade_48k", y, sr=sr3, src_seed=24, src_k=2, src_sel=0, rir_seed=25, rir_len=26000, rir_n=2, rir_sel=0,
        last_rir_sel=1, sample_index=61000, step_time=0.25)
    src48_3 = O.synth_sources(np.random.default_rng(26), sr3, k=1, seconds=3)[0]
    rir48 = O.synth_rir(np.random.default_rng(27), sr3, n=1)
    y, nxt = run_sim(sr3, src48_3, wav_layout(rir48[0]), audio_index=1)
    add("sim48k_multi_i1", y, sr=sr3, src_seed=26, seconds=3, rir_seed=27, rir_n=1, rir_sel=0, rir_len=sr3, audio_index=1,
        next_index=int(nxt))

    # ---- early branch running past the clip end (:433-437): a 3.1-s RIR (irTime allows up to 4 s), index < L, and
    # index + num_sample > len(source): the slice source[:index+num_sample] just ends, i.e. ZEROS past the clip end,
    # not the wrap-around of the steady branch
    rl = O.synth_rir(np.random.default_rng(9), sr, length=50000, n=2)
    rl = rl * np.exp(-np.arange(50000) / 20000.0)[None, None, :].astype(np.float32)
    y = run_cont(sr, src3, wav_layout(rl[0]), 46000)
    add("cont_early_past_end", y, sr=sr, src_seed=1, src_k=3, src_sel=0, rir_seed=9, rir_len=50000, rir_n=2,
        rir_sel=0, rir_decay=20000, sample_index=46000, step_time=0.25)
    # ---- cross-fade whose two RIRs take DIFFERENT branches: current RIR 9000 taps (steady, wraps), previous RIR
    # 50000 taps (early, zeros past the end), both at index 46000
    y = run_cont(sr, src3, wav_layout(rc[0]), 46000, last_rir=wav_layout(rl[1]).astype(np.float64), use_crossfade=True)
    add("cont_crossfade_mixed", y, sr=sr, src_seed=1, src_k=3, src_sel=0, rir_seed=8, rir_len=9000, rir_n=2, rir_sel=0,
        last_rir_seed=9, last_rir_len=50000, last_rir_n=2, last_rir_pick=1, rir_decay_last=20000, sample_index=46000,
        step_time=0.25)

    # ---- savi AudioGoalDataset.compute_audiogoal (random index stubbed)
    class FakeRandom:
        idx = 0
        @staticmethod
        def randint(a, b):
            return FakeRandom.idx
    fw2 = FakeWav()
    savi = load_fn("ss_baselines/savi/pretraining/audiogoal_dataset.py", "AudioGoalDataset",
                   "compute_audiogoal", {"wavfile": fw2, "random": FakeRandom})
    for idx in (0, 2):
        FakeRandom.idx = idx
        s = NS(rir_sampling_rate=sr, source_sound_dict={"s": src5}, audio_length=lambda f: 5)
        fw2.files["r.wav"] = wav_layout(rir[2])
        y = savi(s, "r.wav", "s")
        add(f"savi_i{idx}", y, sr=sr, src_seed=4, seconds=5, rir_seed=2, rir_n=3, rir_sel=2, rir_len=sr,
            audio_index=idx)

    # ---- A7 Intensity
    inten = load_fn("ss_baselines/av_wan/avwan_sensors.py", "Intensity", "get_observation",
                    {"Any": object, "Episode": object})
    s = NS(_sim=NS(get_current_audiogoal_observation=lambda: out["clip1s/audiogoal"]))
    out["clip1s/intensity"] = np.asarray(inten(s, observations=None, episode=None))

    # ---- A3 spectrogram composition (librosa/skimage stubs = oracle restatements)
    #  the day a box has librosa / scikit-image this pins the spectrogram half for real: the REAL libraries
    # are bound when importable, the oracle's restatements otherwise; which one it was is recorded in the npz meta
    stand_ins = {}
    try:
        import librosa as real_librosa
        fake_librosa = real_librosa
        stand_ins["librosa.stft"] = "real librosa " + real_librosa.__version__
    except ImportError:
        fake_librosa = NS(stft=lambda sig, n_fft, hop_length, win_length: O.stft(sig))
        stand_ins["librosa.stft"] = "absent here: oracle.ss_oracle.stft (restated from librosa's documented defaults)"
    try:
        from skimage.measure import block_reduce as real_block_reduce
        block_reduce = real_block_reduce
        stand_ins["skimage.block_reduce"] = "real scikit-image"
    except ImportError:
        block_reduce = lambda a, block_size, func: O.block_reduce_mean(a, block_size)     # noqa: E731
        stand_ins["skimage.block_reduce"] = "absent here: oracle.ss_oracle.block_reduce_mean"
    spec = load_fn("soundspaces/tasks/nav.py", "SpectrogramSensor", "compute_spectrogram",
                   {"librosa": fake_librosa, "block_reduce": block_reduce})
    for name in cases:
        out[name + "/spectrogram"] = spec(out[name + "/audiogoal"]).astype(np.float32)
    out["ones16k/spectrogram_shape"] = np.asarray(spec(np.ones((2, 16000))).shape)
    out["ones44k/spectrogram_shape"] = np.asarray(spec(np.ones((2, 44100))).shape)
    out["ones48k/spectrogram_shape"] = np.asarray(spec(np.ones((2, 48000))).shape)

    # keep the fixture small: full audiogoal for 3 cases, every 5th sample for the rest
    full = {"clip1s", "multi_L1.0_i2", "cont_crossfade"}
    for name in cases:
        a = out[name + "/audiogoal"]
        meta[name]["audiogoal_dtype"] = str(a.dtype)
        if name not in full:
            out[name + "/audiogoal"] = a[:, ::5]
            meta[name]["audiogoal_stride"] = 5
        else:
            meta[name]["audiogoal_stride"] = 1
    out["meta"] = np.frombuffer(json.dumps({"cases": cases, "params": meta, "stand_ins": stand_ins}).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(cases), "cases")


if __name__ == "__main__":
    main()

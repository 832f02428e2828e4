#!/usr/bin/env python
"""Per-kernel timings (HIP events on the launch stream) for the three kernels at a few batch sizes.
usage: python scripts/kbench.py [--sr 16000] [--reps 50] [--sizes 128,2048] [--only fused|conv|spec]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from bench import synth_rir_bank_device
from oracle import ss_oracle as O
from ss_amd import ops
from ss_amd.renderer import BatchedAudioRenderer, RirBank

ap = argparse.ArgumentParser()
ap.add_argument("--sr", type=int, default=16000)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--sizes", default="128,2048")
ap.add_argument("--only", default="")
ap.add_argument("--bank-mib", type=int, default=512)
ap.add_argument("--spectral", action="store_true", help="spectral RIR bank (k_conv_spec) instead of the time-domain bank")
ap.add_argument("--raw", action="store_true", help="launch through bound ctypes calls (host overhead ~3 us per launch)")
ap.add_argument("--sort", action="store_true", help="units sorted by sound id modulo 8 (with SS_HIP_XCD_MAP=1: one group of sounds per XCD)")
ap.add_argument("--distinct", type=int, default=8, help="pre-planned batches cycled (8 x 128 envs fit the Infinity Cache; bench.py streams)")
ap.add_argument("--sounds", type=int, default=102, help="source clips (102 = bench.py: 13 MB of window spectra, more than one XCD's L2)")
a = ap.parse_args()
dev = torch.device("cuda:0")
sr = a.sr
rng = np.random.default_rng(0)
r = BatchedAudioRenderer(sr, device=dev)
for i, c in enumerate(O.synth_sources(rng, sr, k=a.sounds)):
    r.add_source(str(i), c)
R = max(8, (a.bank_mib << 20) // (2 * sr * 4))
r.set_rir_bank(RirBank(synth_rir_bank_device(torch, R, sr, sr, dev, 3), torch.full((R,), sr, dtype=torch.int32, device=dev)))
if a.spectral:
    r.rirs.build_spectra()

def timeit(fn, reps):
    spin_up(fn)
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

_spun = [False]
def spin_up(fn, ms=40.0):
    """GPU clocks take milliseconds to come up (profiles/r2/NOTES.md section 12): run `fn` untimed for `ms` first."""
    import time
    if _spun[0]:
        return
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < ms * 1e-3:
        for _ in range(32):
            fn(k); k += 1
        torch.cuda.synchronize()
    _spun[0] = True

from ss_amd import _lib
LIB = _lib.load()
STREAM = torch.cuda.current_stream().cuda_stream


def raw_conv(plan, out):
    """bound ctypes call (no per-launch torch / Python checks: ~3 us of host time instead of ~11, so that kernels shorter
    than the Python wrapper's overhead are still timed by the events)"""
    if r.rirs.spectra is not None:
        args = (r._spec.data_ptr(), r.rirs.spectra.data_ptr(), r.rirs.lengths.data_ptr(), plan.desc.data_ptr(), out.data_ptr(),
                len(plan), r.rirs.spectra.shape[2], r.n_valid, r.out_len, plan.flags, STREAM)
        fn = LIB.ss_fftconv_binaural_spec_f32
    else:
        cap = r.rirs.cap
        args = (r._spec.data_ptr(), r.rirs.data.data_ptr(), r.rirs.lengths.data_ptr(), plan.desc.data_ptr(), out.data_ptr(),
                len(plan), 2 * cap, cap, 1, cap, r.n_valid, r.out_len, plan.flags, STREAM)
        fn = LIB.ss_fftconv_binaural_f32
    return lambda: fn(*args)


def raw_fused(plan, sg):
    if r.rirs.spectra is not None:
        args = (r._spec.data_ptr(), r.rirs.spectra.data_ptr(), r.rirs.lengths.data_ptr(), plan.desc.data_ptr(), None,
                sg.data_ptr(), len(plan), r.rirs.spectra.shape[2], r.n_valid, r.out_len, 0, plan.flags, STREAM)
        fn = LIB.ss_audio_obs_spec_f32
    else:
        cap = r.rirs.cap
        args = (r._spec.data_ptr(), r.rirs.data.data_ptr(), r.rirs.lengths.data_ptr(), plan.desc.data_ptr(), None,
                sg.data_ptr(), len(plan), 2 * cap, cap, 1, cap, r.n_valid, r.out_len, 0, plan.flags, STREAM)
        fn = LIB.ss_audio_obs_f32
    return lambda: fn(*args)


for N in [int(x) for x in a.sizes.split(",")]:
    def snd():
        s_ = rng.integers(0, a.sounds, N)
        return s_[np.argsort(s_ % 8, kind="stable")] if a.sort else s_
    descs = [r.plan_arrays(snd(), np.zeros(N, np.int64), rng.integers(0, R, N)) for _ in range(a.distinct)]
    ag = torch.empty((N, 2, sr), device=dev); sg = torch.empty((N,) + r.spectrogram_shape, device=dev)
    res = {}
    if a.raw:
        cf = [raw_fused(d, sg) for d in descs]; cc = [raw_conv(d, ag) for d in descs]
        if a.only in ("", "fused"): res["fused"] = timeit(lambda k: cf[k % a.distinct](), a.reps)
        if a.only in ("", "conv"): res["conv"] = timeit(lambda k: cc[k % a.distinct](), a.reps)
        print(f"N={N} sr={sr} raw {'spectral' if a.spectral else 'time'} map={os.environ.get('SS_HIP_XCD_MAP', '0')} sort={int(a.sort)} dbg={os.environ.get('SS_HIP_DBG', '0')} " + " ".join(f"{k}={v:.1f}us" for k, v in res.items()), flush=True)
        continue
    if a.only in ("", "fused"): res["fused"] = timeit(lambda k: r.render(descs[k % a.distinct], spectrogram_out=sg), a.reps)
    if a.only in ("", "conv"): res["conv"] = timeit(lambda k: r.render_audiogoal(descs[k % a.distinct], out=ag), a.reps)
    if a.only in ("", "spec"): res["spec"] = timeit(lambda k: ops.spectrogram_into(ag, sg), a.reps)
    print(f"N={N} sr={sr} {'spectral' if a.spectral else 'time'} map={os.environ.get('SS_HIP_XCD_MAP', '0')} sort={int(a.sort)} " + " ".join(f"{k}={v:.1f}us" for k, v in res.items()), flush=True)

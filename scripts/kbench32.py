#!/usr/bin/env python
"""Same-box A/B of the two FFT cores on the loop-free observation kernels (HIP events on the launch stream):
1024-thread core (k_conv<.., SIMPLE>) against the 512-thread / 32-values-per-thread core (k_conv32), fused and
convolution-only, plus a parity check of one against the other and against the CPU oracle.
usage: python scripts/kbench32.py [--sizes 128,32,512] [--reps 200] [--rounds 3]"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from bench import synth_rir_bank_device
from oracle import ss_oracle as O
from ss_amd import _lib, planning as P
from ss_amd.renderer import BatchedAudioRenderer, RirBank

ap = argparse.ArgumentParser()
ap.add_argument("--sr", type=int, default=16000)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--sizes", default="128,32,512")
ap.add_argument("--bank-mib", type=int, default=1024)
ap.add_argument("--distinct", type=int, default=8)
ap.add_argument("--sounds", type=int, default=102)
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
sr = a.sr
rng = np.random.default_rng(0)
r = BatchedAudioRenderer(sr, device=dev)
clips = O.synth_sources(rng, sr, k=a.sounds)
for i, c in enumerate(clips):
    r.add_source(str(i), c)
R = max(8, (a.bank_mib << 20) // (2 * sr * 4))
bank = synth_rir_bank_device(torch, R, sr, sr, dev, 3)
r.set_rir_bank(RirBank(bank, torch.full((R,), sr, dtype=torch.int32, device=dev)))
LIB = _lib.load()
vp, c_int, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
LIB.ss_source_windows32_f32.argtypes = [vp, vp, vp, c_int, vp]
LIB.ss_audio_obs32_f32.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, vp]
STREAM = torch.cuda.current_stream().cuda_stream


def spec32_of(renderer):
    """the renderer's window spectra again, in the 512-thread core's order (same slots)"""
    s32 = torch.zeros_like(renderer._spec)
    for (sid, t0, wrap), (slot, ws) in renderer._windows.items():
        wd = P.window_desc_rows(ws, renderer.sources.offsets[sid], renderer.sources.lengths[sid], wrap)
        wd_dev = torch.from_numpy(np.ascontiguousarray(wd)).to(dev)
        rc = LIB.ss_source_windows32_f32(renderer.sources.flat().data_ptr(), wd_dev.data_ptr(), s32[slot:slot + len(wd)].data_ptr(),
                                         len(wd), STREAM)
        assert rc == 0, rc
    torch.cuda.synchronize()
    return s32


def timeit(fn, reps):
    for k in range(10): fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def spin(fn, ms=60.0):
    import time
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < ms * 1e-3:
        for _ in range(32):
            fn(k); k += 1
        torch.cuda.synchronize()


results = []
for N in [int(x) for x in a.sizes.split(",")]:
    descs = [r.plan_arrays(rng.integers(0, a.sounds, N), np.zeros(N, np.int64), rng.integers(0, R, N)) for _ in range(a.distinct)]
    s32 = spec32_of(r)
    cap = r.rirs.cap
    ag = torch.empty((N, 2, sr), device=dev); sg = torch.empty((N,) + r.spectrogram_shape, device=dev)
    ag2 = torch.empty_like(ag); sg2 = torch.empty_like(sg)

    def old_fused(d, sgo, ago=None):
        args = (r._spec.data_ptr(), r.rirs.data.data_ptr(), r.rirs.lengths.data_ptr(), d.desc.data_ptr(), ago, sgo.data_ptr(),
                len(d), 2 * cap, cap, 1, cap, r.n_valid, r.out_len, 0, d.flags, STREAM)
        return lambda: LIB.ss_audio_obs_f32(*args)

    def old_conv(d, ago):
        args = (r._spec.data_ptr(), r.rirs.data.data_ptr(), r.rirs.lengths.data_ptr(), d.desc.data_ptr(), ago.data_ptr(),
                len(d), 2 * cap, cap, 1, cap, r.n_valid, r.out_len, d.flags, STREAM)
        return lambda: LIB.ss_fftconv_binaural_f32(*args)

    def new(d, ago, sgo):
        args = (s32.data_ptr(), r.rirs.data.data_ptr(), r.rirs.lengths.data_ptr(), d.desc.data_ptr(),
                ago.data_ptr() if ago is not None else None, sgo.data_ptr() if sgo is not None else None,
                len(d), 2 * cap, cap, 1, cap, r.n_valid, r.out_len, 0, STREAM)
        return lambda: LIB.ss_audio_obs32_f32(*args)

    # parity: new against old (both outputs), and unit 0..3 against the oracle
    ag.zero_(); sg.zero_(); ag2.fill_(7.0); sg2.fill_(7.0)
    assert old_fused(descs[0], sg, ag.data_ptr())() == 0
    assert new(descs[0], ag2, sg2)() == 0
    torch.cuda.synchronize()
    ea = float((ag2 - ag).abs().max() / ag.abs().max()); es = float((sg2 - sg).abs().max() / sg.abs().max())
    sg3 = torch.full_like(sg, 7.0)
    assert new(descs[0], None, sg3)() == 0
    ag3 = torch.full_like(ag, 7.0)
    assert new(descs[0], ag3, None)() == 0
    torch.cuda.synchronize()
    same = bool(torch.equal(sg3, sg2) and torch.equal(ag3, ag2))
    print(f"N={N} parity new-vs-old: audiogoal {ea:.2e} spectrogram {es:.2e}; spectrogram-only / audiogoal-only launches identical: {same}", flush=True)
    assert ea < 1e-5 and es < 1e-5 and same
    of = [old_fused(d, sg) for d in descs]; oc = [old_conv(d, ag) for d in descs]
    nf = [new(d, None, sg2) for d in descs]; nc = [new(d, ag2, None) for d in descs]
    spin(lambda k: of[k % a.distinct]())
    row = {"N": N, "sr": sr, "parity_audiogoal": ea, "parity_spectrogram": es}
    for rd in range(a.rounds):
        for name, fns in (("fused_1024", of), ("fused_512", nf), ("conv_1024", oc), ("conv_512", nc)):
            us = timeit(lambda k: fns[k % a.distinct](), a.reps)
            row.setdefault(name, []).append(round(us, 2))
    print(json.dumps(row), flush=True)
    results.append(row)
if a.out:
    with open(a.out, "w") as f:
        json.dump(results, f, indent=1)

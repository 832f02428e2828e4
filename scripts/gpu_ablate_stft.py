#!/usr/bin/env python
"""Ablation of the stand-alone spectrogram kernel (timing only; results wrong by construction)."""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sound-spaces_amd", "csrc")
TMP = "/tmp/abl/sound-spaces_amd/csrc"
SO = os.path.join(CSRC, "libss_hip.so")
DRY = "--dry" in sys.argv
shutil.copy(SO, "/tmp/base.so")

def sub(text, old, new, must=True):
    assert (old in text) or not must, old
    return text.replace(old, new)

def patched(kind):
    shutil.rmtree("/tmp/abl", ignore_errors=True); shutil.copytree(CSRC, TMP)
    shutil.copytree(os.path.join(ROOT, "include"), "/tmp/abl/include")
    core = open(os.path.join(TMP, "ss_fft_core.hpp")).read()
    kern = open(os.path.join(TMP, "ss_kernels.hpp")).read()
    kinds = kind.split("+")
    if "NOFFT" in kinds:
        for fn in ("void fft16(c32 (&x)[16]) {", "void twiddle16(c32 (&x)[16], c32 w) {"):
            core = sub(core, fn, fn + " return;")
    if "NOMAG" in kinds:
        kern = sub(kern, "        for (int e = 0; e < 4; ++e) {\n            const c32 P = add_conj", "        for (int e = 0; e < 0; ++e) {\n            const c32 P = add_conj")
    if "NOSQRT" in kinds:
        core = sub(core, "    return __builtin_amdgcn_sqrtf(x);", "    return x;")
    if "NOLOG" in kinds:
        kern = sub(kern, "store(r, fast_log1p(v * (1.0f / 16.0f)));", "store(r, v);")
    if "NOSTORE" in kinds:
        kern = sub(kern, "[&](int b, float v) { o[(b * p.t4 + tb4) * 2 + ch] = v; });", '[&](int b, float v) { asm volatile("" :: "v"(v)); });')
    if "NOLOAD" in kinds:
        kern = sub(kern, "            const c32 s = y2[16 * j], w = w2[16 * j];", "            const c32 s = mk2((float)q, (float)j), w = mk2(1.f, 1.f);")
    if "NOTRANSPOSE" in kinds:
        kern = sub(kern, "    for (int r = 0; r < 16; ++r) fr[r * 17 + q] = x[r];", '    for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(x[r]));')
        kern = sub(kern, "    for (int r = 0; r < 16; ++r) x[r] = fr[q * 17 + r];", "    for (int r = 0; r < 16; ++r) x[r] = mk2(x[r].y, x[r].x);")
    if "F_NOFFT" in kinds:      # only the STFT's own FFT work (conv untouched)
        kern = sub(kern, "    fft16<false>(x);\n    SSK_OPAQUE2(wq);\n    twiddle16<false>(x, wq);", "    SSK_OPAQUE2(wq);")
        kern = sub(kern, "    fft16<false>(x);                         // x[s] = Z[q + 16 s]", "")
    if "F_NOSTORE" in kinds:
        kern = sub(kern, "[&](int b, float v) { o[(b * p.t4 + wv) * 2 + ch] = v; });", '[&](int b, float v) { asm volatile("" :: "v"(v)); });')
        kern = sub(kern, "                o[(b * p.t4 + wv + 16) * 2 + ch] = v;", '                asm volatile("" :: "v"(v));')
    if "F_SKIP" in kinds:
        kern = sub(kern, "    if (FUSE) fused_stft_phase(lds, p, t, unit, ch, y, s_win, s_tw512, wq);", "    if (FUSE && p.n_valid < 0) fused_stft_phase(lds, p, t, unit, ch, y, s_win, s_tw512, wq);")
    if "F_ROUND1ONLY" in kinds:
        kern = sub(kern, "    if (p.t4 > 16) {\n        lds_barrier();", "    if (p.t4 > 1600) {\n        lds_barrier();")
    if "EMPTY" in kinds:
        kern = sub(kern, "    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;\n    const int blocks_per_row",
                   "    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;\n    if (p.len > 0) { if (lane == 999) sc[0] = mk2(0.f, 0.f); return; }\n    const int blocks_per_row")
    open(os.path.join(TMP, "ss_fft_core.hpp"), "w").write(core)
    open(os.path.join(TMP, "ss_kernels.hpp"), "w").write(kern)

for kind in ("BASE", "F_SKIP", "F_NOFFT", "NOMAG", "NOTRANSPOSE", "F_NOSTORE", "F_NOFFT+NOMAG+NOTRANSPOSE", "F_ROUND1ONLY", "NOLOAD"):
    patched(kind)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "ss_hip.hip", "-o", "/tmp/abl/out.so" if DRY else SO], cwd=TMP)
    print("== variant", kind, flush=True)
    for _ in range(0 if DRY else 2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kbench.py"), "--sizes", "2048", "--reps", "60",
                              "--only", "fused"], capture_output=True, text=True).stdout
        print(" ".join(l for l in out.splitlines() if l.startswith("N=")), flush=True)
shutil.copy("/tmp/base.so", SO)

#!/usr/bin/env python
"""Ablation builds of the conv kernel k_conv<false,true> (timing only, results are wrong by construction; run with
SS_HIP_NO_ROW_KERNEL=1 semantics: the script sets it, so the 2048-unit launch uses one workgroup per row):
NOVALU = butterflies/twiddles/Hermitian stage skipped (LDS + global traffic + barriers remain)
NOLDS  = LDS reads/writes of the passes removed (VALU + global + barriers remain)
Patches a temporary copy of csrc/, builds it over libss_hip.so, times, restores."""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sound-spaces_amd", "csrc")
TMP = "/tmp/abl/sound-spaces_amd/csrc"
SO = os.path.join(CSRC, "libss_hip.so")
DRY = "--dry" in sys.argv
shutil.copy(SO, "/tmp/base.so")
os.environ["SS_HIP_NO_ROW_KERNEL"] = "1"

def patched(kind):
    shutil.rmtree("/tmp/abl", ignore_errors=True); shutil.copytree(CSRC, TMP)
    shutil.copytree(os.path.join(ROOT, "include"), "/tmp/abl/include")
    core = open(os.path.join(TMP, "ss_fft_core.hpp")).read()
    kern = open(os.path.join(TMP, "ss_kernels.hpp")).read()
    kinds = kind.split("+")
    if "EMPTY" in kinds:
        kern = kern.replace("""    __shared__ c32 lds[kLdsComplex];
    const int t = threadIdx.x;
    const int unit = blockIdx.x >> 1""", """    __shared__ c32 lds[kLdsComplex];
    const int t = threadIdx.x;
    if (p.n_valid >= 0) { if (t == 5000) lds[0] = mk2(0.f, 0.f); return; }
    const int unit = blockIdx.x >> 1""")
    if "NOSPEC" in kinds:      # window-spectrum loads replaced by a constant
        kern = kern.replace("sv0[hh] = sp[hh * 1024];", "sv0[hh] = mk4(mk2(1.f, 0.f), mk2(1.f, 0.f));")
        kern = kern.replace("sv1[hh] = sp[(4 + hh) * 1024];", "sv1[hh] = mk4(mk2(1.f, 0.f), mk2(1.f, 0.f));")
    if "NOSTORE" in kinds:     # output stores kept alive but not issued
        kern = kern.replace("if (t + 1024 * a < m_end) o2[1024 * a] = y[a];", 'asm volatile("" :: "v"(y[a]));')
    if "NORIR" in kinds:       # RIR loads replaced by synthetic values
        kern = kern.replace("return m < m_end ? h2[m] : mk2(0.f, 0.f);", "return mk2((float)m, 1.f);")
    if "NOTW" in kinds:        # per-thread twiddle table loads replaced by constants
        core = core.replace("w.p1 = twM[t];", "w.p1 = mk2(1.f, 0.f);").replace("w.p2 = twM[16 * (t & 63)];", "w.p2 = mk2(1.f, 0.f);")
        core = core.replace("w.i0 = twItem[t];", "w.i0 = mk2(1.f, 0.f);").replace("w.i1 = twItem[t + 1024];", "w.i1 = mk2(1.f, 0.f);")
    if "NOBAR" in kinds:
        core = core.replace("lds_barrier();", "")
        kern = kern.replace("lds_barrier();", "")
    if "NOVALU" in kinds:
        for fn in ("void fft16(c32 (&x)[16]) {", "void fft16_fwd_lo8(c32 (&x)[16]) {", "void twiddle16(c32 (&x)[16], c32 w) {",
                   "void twiddle16_d(c32 (&x)[16], int d_uniform) {", "void herm_fwd(c32& vk, c32& vp, c32 wk) {",
                   "void herm_inv(c32& yk, c32& yp, c32 wk) {", "void bfly4(c32& a, c32& b, c32& c, c32& d) {"):
            assert fn in core, fn
            core = core.replace(fn, fn + " return;")
    if "NOLDS" in kinds:
        # reads -> synthetic values, writes -> keep-alive
        core = re.sub(r"x\[(\w)\] = lds_ld\((base|src) \+ ([^;]+)\);", r"x[\1] = mk2((float)t, (float)\1);", core)
        core = re.sub(r"(base|dst)\[([^\]]+)\] = x\[(\w)\];", r'asm volatile("" :: "v"(x[\3]));', core)
        core = re.sub(r"\{ v\[d\] = lds_ld\(pa \+ 4352 \* d\); v\[4 \+ d\] = lds_ld\(pb \+ 4352 \* d\); \}", r"{ v[d] = mk2((float)q, (float)d); v[4 + d] = mk2((float)d, (float)q); }", core)
        core = re.sub(r"\{ pa\[4352 \* d\] = y\[d\]; pb\[4352 \* d\] = y\[4 \+ d\]; \}", r'{ asm volatile("" :: "v"(y[d]), "v"(y[4 + d])); }', core)
        kern = re.sub(r"for \(int a = 0; a < 16; \+\+a\) base\[1040 \* a\] = x\[a\];", r'for (int a = 0; a < 16; ++a) asm volatile("" :: "v"(x[a]));', kern)
        kern = re.sub(r"for \(int a = 0; a < 16; \+\+a\) x\[a\] = lds_ld\(base \+ 1040 \* a\);", r"for (int a = 0; a < 16; ++a) x[a] = mk2((float)t, (float)a);", kern)
    open(os.path.join(TMP, "ss_fft_core.hpp"), "w").write(core)
    open(os.path.join(TMP, "ss_kernels.hpp"), "w").write(kern)

IO = "NOSPEC+NOSTORE+NORIR+NOTW"
for kind in ("BASE", IO, "NORIR", "NOSTORE", "NOSPEC+NOTW"):
    patched(kind)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "ss_hip.hip", "-o", "/tmp/abl/out.so" if DRY else SO], cwd=TMP)
    print("== variant", kind, flush=True)
    for _ in range(0 if DRY else 2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kbench.py"), "--sizes", "2048", "--reps", "60",
                              "--only", "conv"], capture_output=True, text=True).stdout
        print(" ".join(l for l in out.splitlines() if l.startswith("N=")), flush=True)
shutil.copy("/tmp/base.so", SO)

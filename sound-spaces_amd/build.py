#!/usr/bin/env python
"""Build libss_hip.so (the C-ABI library of include/ss_hip.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to
the GPU box with the repository snapshot.  Usage: python sound-spaces_amd/build.py [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libss_hip.so")
TORCH_SO = os.path.join(CSRC, "libss_torch_ops.so")
SOURCES = ["ss_torch_ops.cpp", "ss_hip.hip", "ss_kernels.hpp", "ss_fft_core.hpp", "ss_kernels32.hpp", "ss_fft_core32.hpp", "ss_features.hpp", "ss_tables.hpp", "ss_context.hpp", "ss_wavio.hpp", os.path.join("..", "..", "include", "ss_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


STAMP = os.path.join(CSRC, ".libss_hip.srchash")
# the device code alone (everything hipcc compiles into libss_hip.so): the key of profiles/rN/traffic.json - PMC counters of a
# kernel stay valid when only the host-side op layer (ss_torch_ops.cpp) changes
KERNEL_SOURCES = [s for s in SOURCES if s != "ss_torch_ops.cpp"]
KERNEL_STAMP = os.path.join(CSRC, ".libss_hip.kernelhash")


def _source_hash(sources=None):
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for s in (SOURCES if sources is None else sources):
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def kernel_hash():
    return _source_hash(KERNEL_SOURCES)


def up_to_date():
    """By content, not mtime: the snapshot that carries the prebuilt library to the GPU box does not keep timestamps."""
    if not (os.path.exists(SO) and os.path.exists(STAMP)):
        return False
    if not os.path.exists(TORCH_SO) and not os.path.exists(TORCH_SO + ".skipped"):
        return False
    return open(STAMP).read().strip() == _source_hash()


def build(force=False, verbose=True):
    build_bench_token(force, verbose)
    if not force and up_to_date():
        if not os.path.exists(KERNEL_STAMP):
            with open(KERNEL_STAMP, "w") as f:
                f.write(kernel_hash() + "\n")
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["ss_hip.hip", "-o", SO]
    if verbose:
        print("[ss_amd] " + " ".join(cmd), flush=True)
    tmp_so = SO + ".tmp"
    subprocess.check_call(cmd[:-1] + [tmp_so], cwd=CSRC)
    guard_isa(hipcc, verbose)
    os.replace(tmp_so, SO)
    try:
        build_torch_ops(verbose)
    except Exception as e:                                      #  a torch with another ABI / no ROCm build must not
        # take libss_hip.so down with it: ss_amd/ops.py keeps its Python registrations of the same ops when the extension is absent
        print(f"[ss_amd] WARNING: libss_torch_ops.so not built ({type(e).__name__}: {e}); torch.ops.ss_hip.* stay Python-registered",
              file=sys.stderr, flush=True)
        if os.path.exists(TORCH_SO):
            os.remove(TORCH_SO)                                 # never pair a stale extension with a new library
        open(TORCH_SO + ".skipped", "w").write(str(e) + "\n")   # (up_to_date(): do not retry on every import)
    with open(STAMP, "w") as f:
        f.write(_source_hash() + "\n")
    with open(KERNEL_STAMP, "w") as f:
        f.write(kernel_hash() + "\n")
    return SO


BENCH_SO = os.path.join(CSRC, "libss_bench.so")
BENCH_STAMP = os.path.join(CSRC, ".libss_bench.srchash")


def build_bench_token(force=False, verbose=True):
    """libss_bench.so: bench.py's stand-in for the policy step between two observations (csrc/ss_bench_token.hip: one
    workgroup reading the slot a step has written).  A library of its own: bench-only code stays out of libss_hip.so."""
    import hashlib
    h = hashlib.sha256(open(os.path.join(CSRC, "ss_bench_token.hip"), "rb").read()).hexdigest()
    if not force and os.path.exists(BENCH_SO) and os.path.exists(BENCH_STAMP) and open(BENCH_STAMP).read().strip() == h:
        return BENCH_SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "ss_bench_token.hip", "-o", BENCH_SO + ".tmp"]
    if verbose:
        print("[ss_amd] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(BENCH_SO + ".tmp", BENCH_SO)
    with open(BENCH_STAMP, "w") as f:
        f.write(h + "\n")
    return BENCH_SO


def build_torch_ops(verbose=True):
    """libss_torch_ops.so: the TORCH_LIBRARY extension (csrc/ss_torch_ops.cpp; host C++ only, calls libss_hip.so through the
    C ABI).  Compiled with g++ against the headers / libraries of the torch that is installed HERE (the GPU box runs the
    same image); finds libss_hip.so next to itself ($ORIGIN)."""
    import torch
    from torch.utils import cpp_extension as ce
    inc = sum((["-I", p] for p in ce.include_paths()), []) + ["-I", "/opt/rocm/include"]
    libdir = ce.library_paths()[0]
    abi = int(bool(getattr(torch._C, "_GLIBCXX_USE_CXX11_ABI", True)))       # the ABI the installed torch was built with
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "ss_torch_ops.cpp", "-o", TORCH_SO + ".tmp",
            "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", f"-D_GLIBCXX_USE_CXX11_ABI={abi}"] + inc +
           ["-L", libdir, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_hip", "-L", CSRC, "-lss_hip",
            "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + libdir])
    if verbose:
        print("[ss_amd] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(TORCH_SO + ".tmp", TORCH_SO)


def guard_isa(hipcc, verbose=True):
    """k_conv_rows issues its RIR prefetch with inline-asm loads the compiler cannot see as asynchronous; check in the
    generated ISA that nothing reads or writes their destination registers before the wait that retires them
    (scripts/check_prefetch_regs.py), and that no kernel spills VGPRs.  A violation fails the build."""
    import re
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "scripts"))
    import check_prefetch_regs
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "ss_hip.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-Wno-unused-command-line-argument", "ss_hip.hip", "-o", asm], cwd=CSRC)
        if check_prefetch_regs.check(asm, verbose=verbose):
            raise RuntimeError("k_conv_rows: prefetch destination registers are touched while the loads are in flight")
        text = open(asm).read()
        spills = [(n, int(v)) for n, v in re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)]
        bad = [(n, v) for n, v in spills if v]
        if bad:
            raise RuntimeError(f"VGPR spills: {bad}")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

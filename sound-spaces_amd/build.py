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
SOURCES = ["ss_hip.hip", "ss_kernels.hpp", "ss_fft_core.hpp", "ss_tables.hpp", os.path.join("..", "..", "include", "ss_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def up_to_date():
    if not os.path.exists(SO):
        return False
    t = os.path.getmtime(SO)
    return all(os.path.getmtime(os.path.join(CSRC, s)) <= t for s in SOURCES)


def build(force=False, verbose=True):
    if not force and up_to_date():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["ss_hip.hip", "-o", SO]
    if verbose:
        print("[ss_amd] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

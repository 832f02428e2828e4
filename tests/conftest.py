import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sound-spaces_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The C-ABI library is a build artefact (git-ignored): make sure it exists and is current before the tests that
    load it run (hipcc cross-compiles gfx950 without a GPU, ~1 min when stale; a no-op otherwise).  Where there is no
    hipcc (nothing to build with) the prebuilt library that travelled with the snapshot is used as is."""
    import importlib.util
    import shutil
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        return
    spec = importlib.util.spec_from_file_location("ss_amd_build", os.path.join(ROOT, "sound-spaces_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(force=False, verbose=False)

#!/usr/bin/env python
"""Registers / LDS / spills of every kernel of libss_hip.so, from the ISA hipcc generates for gfx950 (no GPU needed).
Usage: python scripts/kernel_resources.py [-DSS_AB ...]"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sound-spaces_amd", "csrc")


def main(extra):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "ss_hip.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-Wno-unused-command-line-argument", "ss_hip.hip", "-o", asm] + extra, cwd=CSRC)
        t = open(asm).read()
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:.*\n)*?)\s+\.wavefront_size", t):
        name, body = m.group(1), m.group(2)

        def g(k):
            r = re.search(r"\." + k + r":\s+(\d+)", body)
            return int(r.group(1)) if r else -1
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(f"{dn[:100]:100s} vgpr {g('vgpr_count'):4d} sgpr {g('sgpr_count'):4d} lds {g('group_segment_fixed_size'):7d} "
              f"vspill {g('vgpr_spill_count')} sspill {g('sgpr_spill_count')}")


if __name__ == "__main__":
    main(sys.argv[1:])

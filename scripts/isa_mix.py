#!/usr/bin/env python
"""Static instruction mix of kernels in a hipcc -S listing: python scripts/isa_mix.py /tmp/ss_hip.s name_substring ..."""
import re, sys
from collections import Counter
t = open(sys.argv[1]).read()
def body(name):
    m = re.search(r"^(%s\S*):" % re.escape(name), t, re.M)
    i = m.start(); j = t.index(".Lfunc_end", i)
    return m.group(1), t[i:j]
for n in sys.argv[2:]:
    full, b = body(n)
    ins = [l.strip().split()[0] for l in b.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
    ds = sum(v for k, v in c.items() if k.startswith("ds_"))
    mov = sum(v for k, v in c.items() if k.startswith(("v_mov", "v_cndmask", "v_accvgpr")))
    print(full[:60], "total", len(ins), "valu", valu, "pk", pk, "ds", ds, "mov/cnd", mov, "barrier", c.get("s_barrier", 0), "waitcnt", c.get("s_waitcnt", 0))
    print("   ", c.most_common(16))

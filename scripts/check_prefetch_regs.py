#!/usr/bin/env python
"""Build-time guard for k_conv_rows / k_conv_spec_rows: the registers that the asm prefetch loads write must not be read or written by
any instruction between the loads and the s_waitcnt vmcnt(0) that retires them (the compiler does not know the loads are
asynchronous; a copy or spill in that window would move garbage).  usage: check_prefetch_regs.py file.s"""
import re, sys

def used(l):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', l):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', l):
        out.add(int(m.group(1)))
    return out

def check(path, verbose=True):
    s = open(path).read()
    bad_total = 0
    kernels = [(n, 'global_load_dwordx2', 8) for n in re.findall(r'^(_ZN3ssk11k_conv_rows\w+):', s, re.M)]
    kernels += [(n, 'global_load_dwordx4', 4) for n in re.findall(r'^(_ZN3ssk16k_conv_spec_rows\w+):', s, re.M)]
    for name, op, n_loads in kernels:
        i = s.index(name + ':'); j = s.index('s_endpgm', i)
        L = [l.strip() for l in s[i:j].split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
        loads = [(k, l) for k, l in enumerate(L) if l.startswith(op) and re.search(r', s\[\d+:\d+\]( nt)?$', l)]
        # groups of n_loads consecutive asm loads (prologue, loop); each followed by a bare s_waitcnt vmcnt(0)
        groups, cur = [], []
        for k, l in loads:
            if cur and k - cur[-1][0] > 40:
                groups.append(cur); cur = []
            cur.append((k, l))
        if cur: groups.append(cur)
        for g in groups:
            if len(g) != n_loads:
                continue
            end = next((k for k in range(g[-1][0], len(L)) if L[k].startswith('s_waitcnt vmcnt(0)')), len(L) - 1)
            bad = []
            for k0, l in g:                     # per load: its destination pair from its issue to the retiring wait
                m = re.match(r'global_load_dwordx[24] v\[(\d+):(\d+)\]', l)
                regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
                bad += [(k, L[k]) for k in range(k0 + 1, end) if used(L[k]) & regs]
            if verbose or bad:
                print(f"[isa guard] {name[:44]}: loads at {g[0][0]}..{g[-1][0]}, retired at {end}: {len(bad)} touches")
            for b in bad[:5]: print("   ", b)
            bad_total += len(bad)
    return bad_total

if __name__ == "__main__":
    sys.exit(1 if check(sys.argv[1]) else 0)

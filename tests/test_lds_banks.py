"""Exhaustive bank-conflict check of the FFT core's LDS layouts against the gfx950 rules
(MI355X_MICROARCH: ds_read_b64 = two 32-lane halves over 64 banks; ds_write_b64 = four 16-lane groups over 32 banks).
Mirrors the address expressions of sound-spaces_amd/csrc/ss_fft_core.hpp / ss_kernels.hpp; CPU only.
(Measured on the MI355X: SQ_LDS_BANK_CONFLICT = 4096 cycles per 256-workgroup dispatch of the conv kernel.)"""
import numpy as np


def rd_cost(addrs):
    tot = 0
    for h in range(2):
        banks = {}
        for x in addrs[32 * h:32 * h + 32]:
            for dw in (2 * x, 2 * x + 1):
                banks.setdefault(dw % 64, set()).add(dw)
        tot += max(len(v) for v in banks.values())
    return tot          # 2 = conflict free


def wr_cost(addrs):
    tot = 0
    for g in range(4):
        banks = {}
        for x in addrs[16 * g:16 * g + 16]:
            for dw in (2 * x, 2 * x + 1):
                banks.setdefault(dw % 32, set()).add(dw)
        tot += max(len(v) for v in banks.values())
    return tot          # 4 = conflict free


def pos_a(p):                       # layout A
    return p + (p >> 6)


def pos_b(d, ab, c):                # layout B
    return d * 4352 + ab * 17 + c


def item_gA(q):
    c, lo = q & 15, q >> 4
    if lo:
        return lo + 256 * c
    return 256 * c if c < 8 else 128 + 256 * (c - 8)


def gaddr(g, d):
    return pos_b(d, ((g & 15) << 4) | ((g >> 4) & 15), g >> 8)


def test_item_groups_partition_all_radix4_groups():
    seen = set()
    for q in range(2048):
        gA = item_gA(q)
        gB = 2048 if q == 0 else 4096 - gA
        assert gA not in seen and gB not in seen
        seen |= {gA, gB}
    assert len(seen) == 4096


def test_every_pass_is_bank_conflict_free():
    worst = {}

    def upd(k, v):
        worst[k] = max(worst.get(k, 0), v)
    for wave in range(16):
        t = np.arange(64) + 64 * wave
        for r in range(16):
            ad = [pos_a(int(x) + 1024 * r) for x in t]                                  # pass 1 write / pass 1' read
            upd("p1_write", wr_cost(ad)); upd("p1inv_read", rd_cost(ad))
            ad = [pos_a((int(x) >> 6) * 1024 + (int(x) & 63) + 64 * r) for x in t]      # pass 2 in place
            upd("p2_read", rd_cost(ad)); upd("p2_write", wr_cost(ad))
            a3 = [pos_a((int(x) & 255) * 64 + (int(x) >> 8) + 4 * r) for x in t]        # pass 3: thread = d*256 + ab
            b3 = [pos_b(int(x) >> 8, int(x) & 255, r) for x in t]
            upd("p3_read_A", rd_cost(a3)); upd("p3inv_write_A", wr_cost(a3))
            upd("p3_write_B", wr_cost(b3)); upd("p3inv_read_B", rd_cost(b3))
    assert all(v == 2 for k, v in worst.items() if "read" in k), worst
    assert all(v == 4 for k, v in worst.items() if "write" in k), worst


def test_item_stage_accesses():
    costs_r, costs_w = [], []
    for wave in range(16):
        for s in range(2):
            q = np.arange(64) + 64 * wave + 1024 * s
            gA = [item_gA(int(x)) for x in q]
            gB = [2048 if int(x) == 0 else 4096 - g for x, g in zip(q, gA)]
            for d in range(4):
                for grp in (gA, gB):
                    ad = [gaddr(g, d) for g in grp]
                    costs_r.append(rd_cost(ad))
                    costs_w.append(wr_cost(ad))
    # conflict free except for the handful of irregular lanes of the first wave
    assert np.mean(costs_r) <= 2.2 and np.mean(costs_w) <= 4.2, (np.mean(costs_r), np.mean(costs_w))

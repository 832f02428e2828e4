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


# ---- STFT stage (ss_kernels.hpp: stft_block) --------------------------------------------------------------------
# ds_read_b128 is serviced in four groups of 16 lanes (MI355X_MICROARCH, LDS table), 64 banks of 4 B
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
FRAME_STRIDE, NAT_STRIDE = 272, 288


def pos_n(k):
    return k + 2 * (k >> 5)


def rd128_cost(dw):                  # dw[lane] = dword address of a 16-byte read
    tot = 0
    for g in B128_GROUPS:
        banks = {}
        for l in g:
            for d in range(4):
                banks.setdefault((dw[l] + d) % 64, set()).add(dw[l] + d)
        tot += max(len(v) for v in banks.values())
    return tot          # 4 = conflict free


def test_stft_stage_accesses_are_conflict_free():
    lanes = [(l >> 4, l & 15) for l in range(64)]
    for r in range(16):
        assert wr_cost([f * FRAME_STRIDE + r * 17 + q for f, q in lanes]) == 4          # transpose write
        assert rd_cost([f * FRAME_STRIDE + q * 17 + r for f, q in lanes]) == 2          # transpose read
        assert wr_cost([f * NAT_STRIDE + q + pos_n(16 * r) for f, q in lanes]) == 4     # natural-order write
    for i in range(2):
        for off in (0, 4):           # the two 16-byte halves of Z[4b..4b+3]
            zk = [2 * (f * NAT_STRIDE + pos_n(4 * (q + 16 * i))) + off for f, q in lanes]
            zp = [2 * (f * NAT_STRIDE + pos_n(252 - 4 * (q + 16 * i))) + off for f, q in lanes]
            tw = [2 * pos_n(4 * (q + 16 * i)) + off for f, q in lanes]
            assert rd128_cost(zk) == 4 and rd128_cost(zp) == 4 and rd128_cost(tw) == 4
    # the unpadded layout this replaced was 2-way conflicted on those reads
    bad = [2 * (f * 272 + 4 * q) for f, q in lanes]
    assert rd128_cost(bad) > 4
    assert max(pos_n(255) + 1, 15 * 17 + 16) <= NAT_STRIDE and pos_n(255) < 270


# ---- k_features (ss_features.hpp): the PHAT cross-spectrum V and the lag rows ------------------------------------------
def pos_v(k):
    return (k & 3) * 68 + (k >> 2)


def wr32_cost(dw):                   # ds_write_b32: two groups of 32 lanes over 32 banks (2 = conflict free)
    tot = 0
    for h in range(2):
        banks = {}
        for x in dw[32 * h:32 * h + 32]:
            banks.setdefault(x % 32, set()).add(x)
        tot += max(len(v) for v in banks.values())
    return tot


def test_features_cross_spectrum_layout_and_lag_rows_are_conflict_free():
    V_STRIDE, RES_STRIDE = 272, 17
    lanes = [(l >> 4, l & 15) for l in range(64)]
    assert sorted(pos_v(k) for k in range(256)) == sorted(set(pos_v(k) for k in range(256))) and pos_v(255) < V_STRIDE
    for i in range(2):
        for e in range(4):
            assert wr_cost([f * V_STRIDE + pos_v(4 * (q + 16 * i) + e) for f, q in lanes]) == 4
            mirror = [f * V_STRIDE + pos_v((256 - 4 * (q + 16 * i) - e) % 256) for f, q in lanes]     # (k = 0 is not stored)
            assert wr_cost(mirror) == 4
    for j in range(16):
        assert rd_cost([f * V_STRIDE + pos_v(q + 16 * j) for f, q in lanes]) == 2
    # the layout this replaced (natural order + posN padding, frames 288 apart)
    assert wr_cost([f * NAT_STRIDE + pos_n(4 * q) for f, q in lanes]) > 4
    # lag rows: lane (f, q) stores lag 2 q + u of frame f; at a 16-float pitch the sixteen q of a frame shared one bank
    for u in range(2):
        assert wr32_cost([(2 * q + u + 32) * RES_STRIDE + f for f, q in lanes]) <= 4       # (2-way costs a b32 store nothing)
        assert wr32_cost([(2 * q + u + 32) * 16 + f for f, q in lanes]) == 32

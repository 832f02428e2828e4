// ss_fft_core32.hpp — the 512-thread / 32-values-per-thread form of the 16384-point block FFT (gfx950).
//
// Round 4.  ss_fft_core.hpp runs a block FFT as 16*16*16*4 on 1024 threads: four passes, three LDS exchanges and
// four workgroup barriers per transform, 128 VGPRs per thread.  The measurements of round 3 (profiles/r3/NOTES.md
// section 7) said the kernels are bound by that structure: VALU 64 % + LDS 34 % of a wave's life, serialised at every
// phase boundary, a quarter of the VALU instructions non-arithmetic (addresses rematerialised at the 128-VGPR cap).
// This core factors 16384 = 32 * 32 * 16 on 512 threads (8 waves, 256 VGPRs each):
//
//   pass 1   radix-32 over a   (n = 512 a + 16 b + c)      global/registers -> LDS     twiddle W_16384^{t k1}
//   pass 2   radix-32 over b   in place                                               twiddle W_512^{c k2} (LDS table)
//   item     radix-16 over c   in place, on the TWO groups g and 1024-g whose bins k = g + 1024 k3 are each other's
//            Hermitian partners: split, multiply by the window spectrum, merge, inverse radix-16 - all in registers
//   pass 2', pass 1'           the conjugate passes
//
// i.e. TWO exchanges and two barriers per transform (4 + 4 per convolution instead of 6 + 8), one twiddle stage
// less, half the per-point address arithmetic, and ONE LDS layout that every pass reads and writes in place without
// bank conflicts:
//
//   pos(k1, b, c) = 513 k1 + 16 b + c          (32 * 513 = 16416 complex = 131 328 B)
//
//   pass 1 writes / pass 1' reads:  thread t = 16 b + c, fixed k1: 513 k1 + t      contiguous lanes
//   pass 2 (both ways):             thread = k1 + 32 c, fixed b:   k1 + 16 b + c   (mod 32) distinct over k1
//   item stage:                     thread q -> group q: k1 = q & 31, k2 = q >> 5, fixed c: same rule
//   (tests/test_lds_banks.py enumerates all of them against the ds_read_b64 / ds_write_b64 rules.)
//
// Spectra produced / consumed by this core are in ITS register order (thread q, slot e: e < 16 -> bin q + 1024 e,
// e >= 16 -> bin (1024 - q) + 1024 (e - 16); thread 0: groups 0 and 512), stored as f32x4 i (slots 2i, 2i+1) at
// [i * 512 + q]: 16 coalesced 16-byte loads per lane.  Same size as the 1024-thread order (8192 f32x4), different
// permutation: a spectrum is only ever read by the core that wrote it.
#pragma once
#include "ss_fft_core.hpp"

namespace ssk {

constexpr int kT32 = 512;                     // threads per workgroup of this core
constexpr int kLds32Complex = 32 * 513;       // 16416 complex
constexpr int kTwP2 = 512;                    // entries of the pass-2 twiddle table exp(-2 pi i c k2 / 512) at [c * 32 + k2]

// -i (a - b) and +i (a - b): the W32^8 twiddle of the radix-2 split folded into its subtraction
SSK_PK2(sub_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]", mk2(a.y - b.y, b.x - a.x))
SSK_PK2(sub_pi, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,1]", mk2(b.y - a.y, a.x - b.x))

// exp(-+2 pi i j / 32) as literals (kCos64 / kSin64 are the 64th roots)
template <bool INV, int J>
__device__ __forceinline__ c32 mul_w32(c32 v) {
    if (J == 0) return v;
    return cmul_k(v, kCos64[2 * J], (INV ? 1.f : -1.f) * kSin64[2 * J]);
}

// 32-point DFT, natural-order result in x.  Forward: decimation in frequency (radix-2 split on the inputs, two
// 16-point DFTs); HALF: x[16..31] are known to be zero (zero-padded RIR block).  189 / 221 packed instructions.
template <bool HALF>
__device__ __forceinline__ void fft32_fwd(c32 (&x)[32]) {
    c32 u[16], v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (HALF) { u[j] = x[j]; v[j] = x[j]; }
        else { u[j] = cadd(x[j], x[j + 16]); v[j] = (j == 8) ? sub_mi(x[j], x[j + 16]) : csub(x[j], x[j + 16]); }
    }
    v[1] = mul_w32<false, 1>(v[1]);    v[2] = mul_w32<false, 2>(v[2]);    v[3] = mul_w32<false, 3>(v[3]);
    v[4] = mul_w32<false, 4>(v[4]);    v[5] = mul_w32<false, 5>(v[5]);    v[6] = mul_w32<false, 6>(v[6]);
    v[7] = mul_w32<false, 7>(v[7]);
    if (HALF) v[8] = cmul_k(v[8], 0.f, -1.f);
    v[9] = mul_w32<false, 9>(v[9]);    v[10] = mul_w32<false, 10>(v[10]); v[11] = mul_w32<false, 11>(v[11]);
    v[12] = mul_w32<false, 12>(v[12]); v[13] = mul_w32<false, 13>(v[13]); v[14] = mul_w32<false, 14>(v[14]);
    v[15] = mul_w32<false, 15>(v[15]);
    fft16<false>(u);
    fft16<false>(v);
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[2 * r] = u[r]; x[2 * r + 1] = v[r]; }
}

// Inverse 32-point DFT (conjugate kernel, no scaling): decimation in time, the mirror image of the forward one.
// HI_ONLY: only outputs 16..31 are produced (x[16..31]; the alias-free half of the circular convolution).
template <bool HI_ONLY>
__device__ __forceinline__ void fft32_inv(c32 (&x)[32]) {
    c32 e[16], o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { e[r] = x[2 * r]; o[r] = x[2 * r + 1]; }
    fft16<true>(e);
    fft16<true>(o);
    o[1] = mul_w32<true, 1>(o[1]);    o[2] = mul_w32<true, 2>(o[2]);    o[3] = mul_w32<true, 3>(o[3]);
    o[4] = mul_w32<true, 4>(o[4]);    o[5] = mul_w32<true, 5>(o[5]);    o[6] = mul_w32<true, 6>(o[6]);
    o[7] = mul_w32<true, 7>(o[7]);
    o[9] = mul_w32<true, 9>(o[9]);    o[10] = mul_w32<true, 10>(o[10]); o[11] = mul_w32<true, 11>(o[11]);
    o[12] = mul_w32<true, 12>(o[12]); o[13] = mul_w32<true, 13>(o[13]); o[14] = mul_w32<true, 14>(o[14]);
    o[15] = mul_w32<true, 15>(o[15]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r == 8) {                                  // o[8] carries the twiddle +i: folded into the add / sub
            if (!HI_ONLY) x[r] = add_pi(e[r], o[r]);
            x[r + 16] = add_mi(e[r], o[r]);
        } else {
            if (!HI_ONLY) x[r] = cadd(e[r], o[r]);
            x[r + 16] = csub(e[r], o[r]);
        }
    }
}

// x[r] *= w^r (INV: conj(w)^r), r = 1..31; the powers by a product chain of depth <= 5.  122 packed instructions.
template <bool INV>
__device__ __forceinline__ void twiddle32(c32 (&x)[32], c32 w) {
    c32 p[32];
    p[1] = w;
    p[2] = cmul(w, w);
    p[3] = cmul(p[2], w);      p[4] = cmul(p[2], p[2]);
#pragma unroll
    for (int r = 1; r <= 4; ++r) x[r] = cmul_dir<INV>(x[r], p[r]);
    p[5] = cmul(p[4], p[1]);   p[6] = cmul(p[4], p[2]);   p[7] = cmul(p[4], p[3]);   p[8] = cmul(p[4], p[4]);
#pragma unroll
    for (int r = 5; r <= 8; ++r) x[r] = cmul_dir<INV>(x[r], p[r]);
#pragma unroll
    for (int r = 9; r <= 16; ++r) { p[r] = cmul(p[8], p[r - 8]); x[r] = cmul_dir<INV>(x[r], p[r]); }
#pragma unroll
    for (int r = 17; r <= 31; ++r) x[r] = cmul_dir<INV>(x[r], cmul(p[16], p[r - 16]));
}

__device__ __forceinline__ int pos32(int k1, int b, int c) { return 513 * k1 + 16 * b + c; }

// Per-thread base twiddles of this core: one table entry each, loaded once at kernel start.
struct ThreadTw32 {
    c32 p1;      // exp(-2 pi i t / 16384)            passes 1 / 1'
    c32 g;       // exp(-2 pi i t / 32768)            Hermitian stage of item t (group t)
};
__device__ __forceinline__ ThreadTw32 load_thread_tw32(const c32* __restrict__ twM, const c32* __restrict__ twG, int t) {
    ThreadTw32 w;
    w.p1 = twM[t];
    w.g = twG[t];
    return w;
}

// pass 1 forward (registers -> LDS).  LOADER(m) returns the packed sample pair (x[2m], x[2m+1]), m = t + 512 a.
// HALF: packed samples m >= 8192 are known to be zero.
template <bool HALF, class LOADER>
__device__ __forceinline__ void pass1_fwd32(c32* lds, c32 wbase, int t, LOADER load) {
    c32 x[32];
#pragma unroll
    for (int a = 0; a < 32; ++a) x[a] = (HALF && a >= 16) ? mk2(0.f, 0.f) : load(t + 512 * a);
    fft32_fwd<HALF>(x);
    c32 w = wbase;
    SSK_OPAQUE2(w);
    twiddle32<false>(x, w);
    c32* base = lds + t;
#pragma unroll
    for (int k = 0; k < 32; ++k) lds_st(base + 513 * k, x[k]);
}

// pass 1 inverse: LDS -> registers, upper half only: y[j] <-> packed sample m = t + 512 (16 + j)
__device__ __forceinline__ void pass1_inv32(const c32* lds, c32 wbase, int t, c32 (&y)[16]) {
    c32 x[32];
    const c32* base = lds + t;
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k] = lds_ld(base + 513 * k);
    c32 w = wbase;
    SSK_OPAQUE2(w);
    twiddle32<true>(x, w);
    fft32_inv<true>(x);
#pragma unroll
    for (int j = 0; j < 16; ++j) y[j] = x[16 + j];
}

// pass 2, in place: thread = k1 + 32 c; tw2 = the LDS copy of exp(-2 pi i c k2 / 512) at [c * 32 + k2] (16-byte aligned)
template <bool INV>
__device__ __forceinline__ void pass2_32(c32* lds, const c32* tw2, int t) {
    const int k1 = t & 31, c = t >> 5;
    c32* base = lds + 513 * k1 + c;
    const f32x4* w4 = reinterpret_cast<const f32x4*>(tw2 + 32 * c);
    c32 x[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) x[b] = lds_ld(base + 16 * b);
    if (INV) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 w = w4[i];
            if (i) x[2 * i] = cmulc(x[2 * i], w.xy);
            x[2 * i + 1] = cmulc(x[2 * i + 1], w.zw);
        }
        fft32_inv<false>(x);
    } else {
        fft32_fwd<false>(x);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 w = w4[i];
            if (i) x[2 * i] = cmul(x[2 * i], w.xy);
            x[2 * i + 1] = cmul(x[2 * i + 1], w.zw);
        }
    }
#pragma unroll
    for (int b = 0; b < 32; ++b) lds_st(base + 16 * b, x[b]);
}

// exp(-2 pi i j / 32) * wg and exp(-2 pi i (2 j + 1) / 64) as literals for the Hermitian stage
template <int J>
__device__ __forceinline__ c32 herm_w(c32 wg) {
    if (J == 0) return wg;
    if (J == 8) return mk2(wg.y, -wg.x);
    return cmul_k(wg, kCos64[2 * J], -kSin64[2 * J]);
}

// LDS addresses of the two groups of item q (thread q): gA = q, gB = 1024 - q; item 0: groups 0 and 512
__device__ __forceinline__ void item32_ptrs(c32* lds, int q, c32*& pa, c32*& pb) {
    const int gB = (q == 0) ? 512 : 1024 - q;
    pa = lds + pos32(q & 31, q >> 5, 0);
    pb = lds + pos32(gB & 31, gB >> 5, 0);
}

// Item stage, forward half: the two radix-16 groups of item q -> 32 bins of 2*rFFT_32768 in slot order
//   v[e]      = X2[q + 1024 e]            e < 16
//   v[16 + e] = X2[(1024 - q) + 1024 e]
// (item 0: v[0] = (X2[0], X2[16384]), both real; v[e] = X2[1024 e]; v[16 + e] = X2[512 + 1024 e])
__device__ __forceinline__ void item32_herm_fwd(c32 wbase, int q, c32 (&v)[32]) {
    c32 wg = wbase;                     // exp(-2 pi i q / 32768)
    SSK_OPAQUE2(wg);
    if (q != 0) {
        // bin k = q + 1024 e pairs with 16384 - k = (1024 - q) + 1024 (15 - e): slots e and 31 - e
        herm_fwd(v[0], v[31], herm_w<0>(wg));    herm_fwd(v[1], v[30], herm_w<1>(wg));
        herm_fwd(v[2], v[29], herm_w<2>(wg));    herm_fwd(v[3], v[28], herm_w<3>(wg));
        herm_fwd(v[4], v[27], herm_w<4>(wg));    herm_fwd(v[5], v[26], herm_w<5>(wg));
        herm_fwd(v[6], v[25], herm_w<6>(wg));    herm_fwd(v[7], v[24], herm_w<7>(wg));
        herm_fwd(v[8], v[23], herm_w<8>(wg));    herm_fwd(v[9], v[22], herm_w<9>(wg));
        herm_fwd(v[10], v[21], herm_w<10>(wg));  herm_fwd(v[11], v[20], herm_w<11>(wg));
        herm_fwd(v[12], v[19], herm_w<12>(wg));  herm_fwd(v[13], v[18], herm_w<13>(wg));
        herm_fwd(v[14], v[17], herm_w<14>(wg));  herm_fwd(v[15], v[16], herm_w<15>(wg));
    } else {
        const c32 v0 = v[0];
        v[0] = mk2(2.f * (v0.x + v0.y), 2.f * (v0.x - v0.y));      // X2[0], X2[16384]
        // group 0: k = 1024 e pairs with 1024 (16 - e); wk = exp(-2 pi i e / 32)
        herm_fwd(v[1], v[15], mk2(kCos64[2], -kSin64[2]));     herm_fwd(v[2], v[14], mk2(kCos64[4], -kSin64[4]));
        herm_fwd(v[3], v[13], mk2(kCos64[6], -kSin64[6]));     herm_fwd(v[4], v[12], mk2(kCos64[8], -kSin64[8]));
        herm_fwd(v[5], v[11], mk2(kCos64[10], -kSin64[10]));   herm_fwd(v[6], v[10], mk2(kCos64[12], -kSin64[12]));
        herm_fwd(v[7], v[9], mk2(kCos64[14], -kSin64[14]));
        c32 dup = v[8]; herm_fwd(v[8], dup, mk2(0.f, -1.f));       // k = 8192 (self)
        // group 512: k = 512 + 1024 e pairs with 512 + 1024 (15 - e); wk = exp(-2 pi i (2 e + 1) / 64)
        herm_fwd(v[16], v[31], mk2(kCos64[1], -kSin64[1]));    herm_fwd(v[17], v[30], mk2(kCos64[3], -kSin64[3]));
        herm_fwd(v[18], v[29], mk2(kCos64[5], -kSin64[5]));    herm_fwd(v[19], v[28], mk2(kCos64[7], -kSin64[7]));
        herm_fwd(v[20], v[27], mk2(kCos64[9], -kSin64[9]));    herm_fwd(v[21], v[26], mk2(kCos64[11], -kSin64[11]));
        herm_fwd(v[22], v[25], mk2(kCos64[13], -kSin64[13]));  herm_fwd(v[23], v[24], mk2(kCos64[15], -kSin64[15]));
    }
}

// Item stage, inverse half: Y2 bins in slot order -> the packed spectrum V'2 of the real output (both groups)
__device__ __forceinline__ void item32_herm_inv(c32 wbase, int q, c32 (&y)[32]) {
    c32 wg = wbase;
    SSK_OPAQUE2(wg);
    if (q != 0) {
        herm_inv(y[0], y[31], herm_w<0>(wg));    herm_inv(y[1], y[30], herm_w<1>(wg));
        herm_inv(y[2], y[29], herm_w<2>(wg));    herm_inv(y[3], y[28], herm_w<3>(wg));
        herm_inv(y[4], y[27], herm_w<4>(wg));    herm_inv(y[5], y[26], herm_w<5>(wg));
        herm_inv(y[6], y[25], herm_w<6>(wg));    herm_inv(y[7], y[24], herm_w<7>(wg));
        herm_inv(y[8], y[23], herm_w<8>(wg));    herm_inv(y[9], y[22], herm_w<9>(wg));
        herm_inv(y[10], y[21], herm_w<10>(wg));  herm_inv(y[11], y[20], herm_w<11>(wg));
        herm_inv(y[12], y[19], herm_w<12>(wg));  herm_inv(y[13], y[18], herm_w<13>(wg));
        herm_inv(y[14], y[17], herm_w<14>(wg));  herm_inv(y[15], y[16], herm_w<15>(wg));
    } else {
        const c32 y0 = y[0];                                            // (Y2[0], Y2[16384])
        y[0] = mk2(y0.x + y0.y, y0.x - y0.y);
        herm_inv(y[1], y[15], mk2(kCos64[2], -kSin64[2]));     herm_inv(y[2], y[14], mk2(kCos64[4], -kSin64[4]));
        herm_inv(y[3], y[13], mk2(kCos64[6], -kSin64[6]));     herm_inv(y[4], y[12], mk2(kCos64[8], -kSin64[8]));
        herm_inv(y[5], y[11], mk2(kCos64[10], -kSin64[10]));   herm_inv(y[6], y[10], mk2(kCos64[12], -kSin64[12]));
        herm_inv(y[7], y[9], mk2(kCos64[14], -kSin64[14]));
        c32 dup = y[8]; herm_inv(y[8], dup, mk2(0.f, -1.f));
        herm_inv(y[16], y[31], mk2(kCos64[1], -kSin64[1]));    herm_inv(y[17], y[30], mk2(kCos64[3], -kSin64[3]));
        herm_inv(y[18], y[29], mk2(kCos64[5], -kSin64[5]));    herm_inv(y[19], y[28], mk2(kCos64[7], -kSin64[7]));
        herm_inv(y[20], y[27], mk2(kCos64[9], -kSin64[9]));    herm_inv(y[21], y[26], mk2(kCos64[11], -kSin64[11]));
        herm_inv(y[22], y[25], mk2(kCos64[13], -kSin64[13]));  herm_inv(y[23], y[24], mk2(kCos64[15], -kSin64[15]));
    }
}

// load the two groups of item q from LDS and finish the 16384-point FFT (radix-16 over c), then the Hermitian split
__device__ __forceinline__ void item32_load_fwd(c32* lds, c32 wbase, int q, c32 (&v)[32]) {
    c32 *pa, *pb;
    item32_ptrs(lds, q, pa, pb);
    c32 a[16], b[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { a[c] = lds_ld(pa + c); b[c] = lds_ld(pb + c); }
    fft16<false>(a);
    fft16<false>(b);
#pragma unroll
    for (int e = 0; e < 16; ++e) { v[e] = a[e]; v[16 + e] = b[e]; }
    item32_herm_fwd(wbase, q, v);
}

// the inverse: merge, inverse radix-16 on both groups, back to the same LDS slots
__device__ __forceinline__ void item32_store_inv(c32* lds, c32 wbase, int q, c32 (&y)[32]) {
    c32 *pa, *pb;
    item32_ptrs(lds, q, pa, pb);
    item32_herm_inv(wbase, q, y);
    c32 a[16], b[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { a[e] = y[e]; b[e] = y[16 + e]; }
    fft16<true>(a);
    fft16<true>(b);
#pragma unroll
    for (int c = 0; c < 16; ++c) { lds_st(pa + c, a[c]); lds_st(pb + c, b[c]); }
}

}  // namespace ssk

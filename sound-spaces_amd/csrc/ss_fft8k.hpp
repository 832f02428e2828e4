// ss_fft8k.hpp — 8192-point complex FFT building blocks for HALF-ROW workgroups (gfx950).
//
// Why a second FFT size.  The 16384-point core (ss_fft_core.hpp) needs 136 KiB of LDS, so ONE 1024-thread workgroup
// fits a CU and its phases - RIR load, 7 FFT passes, spectral multiply, stores / STFT - run strictly one after the
// other: profiles/r1 and r2 show the kernel at ~45 % of its own instruction-issue time, the rest being exposed memory
// latency, barrier skew and LDS phases that nothing overlaps.  Here a workgroup is 512 threads (8 wave64) around an
// 8192-point FFT = 68 KiB of LDS, two workgroups share a CU, and while one waits for HBM or sits in an LDS phase the
// other one issues arithmetic.
//
//   8192 = 16 * 16 * 16 * 2:  n = 512 a + 32 b + 2 c + d,   k = a' + 16 b' + 256 c' + 4096 d'
//   pass 1: a -> a', twiddle W8192^(a' (n mod 512))       thread t = n mod 512
//   pass 2: b -> b', twiddle W512^(b' (2c + d))           thread (a' = t >> 5, low = t & 31 = 2c + d)
//   pass 3: c -> c', twiddle W32^(c' d)                   thread (d = t >> 8, ab = 16 a' + b' = t & 255)
//   items : d -> d' (radix 2) fused with the Hermitian split of the packed real transform
// The group index g = a' + 16 b' + 256 c' has the SAME 4096 values and the same Hermitian pairing g <-> 4096 - g as the
// radix-4 groups of the 16384-point core, so posB / group_ab / group_c / item_gA are shared; only the last radix
// (2 instead of 4: an item = 2 groups x 2 bins, 4 items per thread) and the strides differ.
// LDS layouts (ds_read/write_b64 conflict free, tests/test_lds_banks.py):
//   layout A8 (passes 1-2-3): posA8(p) = p + (p >> 5)         (1 complex of padding per 32; pass 3 reads it at lane stride 33)
//   layout B  (pass 3 - items): posB(d, ab, c) = d*4352 + ab*17 + c, d < 2
#pragma once
#include "ss_fft_core.hpp"

namespace ssk8 {

using ssk::c32;
using ssk::f32x4;
using ssk::mk2;
using ssk::lds_ld;
using ssk::lds_st;
using ssk::lds_barrier;
using ssk::fft16;
using ssk::fft16_fwd_lo8;
using ssk::twiddle16;
using ssk::cmul;
using ssk::cmulc;
using ssk::cmul_k;
using ssk::cadd;
using ssk::csub;
using ssk::herm_fwd;
using ssk::herm_inv;
using ssk::posB;
using ssk::group_ab;
using ssk::group_c;
using ssk::item_gA;

constexpr int kM8 = 8192;              // complex points
constexpr int kT8 = 512;               // threads per workgroup
constexpr int kSeg = 16384;            // real samples one FFT covers
constexpr int kP = 8000;               // RIR partition length (taps) = hop between the windows of consecutive partitions
constexpr int kValid = kSeg - kP + 1;  // 8385 alias-free output samples per block: circular indices kP-1 .. kSeg-1
constexpr int kLds8 = 2 * 4352;        // 8704 complex = 69632 B (layout B; layout A8 needs 8448)
constexpr int kSpec8 = 8192;           // complex values of one stored window / block spectrum (kernel order)
constexpr float kWindowScale8 = 1.0f / (8.0f * 8192.0f);   // (2X -> X) * 1/(4M)

__device__ __forceinline__ int posA8(int p) { return p + (p >> 5); }

// per-thread base twiddles, loaded once (see ssk::ThreadTw)
struct ThreadTw8 {
    c32 p1;          // exp(-2 pi i t / 8192)                 passes 1 / 1'
    c32 p2;          // exp(-2 pi i 16 (t & 31) / 8192)       passes 2 / 2'
    c32 it[4];       // exp(-2 pi i gA(t + 512 s) / 16384)    Hermitian stage of the four items
};
// twM = exp(-2 pi i t / 16384), t < 1024 (the 16384-point table): entry 2t is exp(-2 pi i t / 8192)
__device__ __forceinline__ ThreadTw8 load_thread_tw8(const c32* __restrict__ twM, const c32* __restrict__ twItem8, int t) {
    ThreadTw8 w;
    w.p1 = twM[2 * t];
    w.p2 = twM[32 * (t & 31)];
#pragma unroll
    for (int s = 0; s < 4; ++s) w.it[s] = twItem8[t + 512 * s];
    return w;
}

// ---- pass 1 -------------------------------------------------------------------------------------------------
// forward: LOADER(m) = packed sample pair (x[2m], x[2m+1]); HALF: m >= 4096 known zero (an RIR partition has <= 8192 taps)
template <bool HALF, class LOADER>
__device__ __forceinline__ void pass1_fwd8(c32* lds, c32 wbase, int t, LOADER load) {
    c32 x[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) x[a] = (HALF && a >= 8) ? mk2(0.f, 0.f) : load(t + 512 * a);
    if (HALF) fft16_fwd_lo8(x); else fft16<false>(x);
    c32 w = wbase;
    SSK_OPAQUE2(w);
    twiddle16<false>(x, w);
    c32* base = lds + t + (t >> 5);             // posA8(t + 512 a') = t + (t >> 5) + 528 a'
#pragma unroll
    for (int a = 0; a < 16; ++a) lds_st(base + 528 * a, x[a]);
}
// inverse: all 16 outputs y[a] <-> packed pair m = t + 512 a (the caller keeps the ones it needs)
__device__ __forceinline__ void pass1_inv8(const c32* lds, c32 wbase, int t, c32 (&x)[16]) {
    const c32* base = lds + t + (t >> 5);
#pragma unroll
    for (int a = 0; a < 16; ++a) x[a] = lds_ld(base + 528 * a);
    c32 w = wbase;
    SSK_OPAQUE2(w);
    twiddle16<true>(x, w);
    fft16<true>(x);
}

// ---- pass 2 (in place, layout A8): a' = t >> 5, low = t & 31 --------------------------------------------------
template <bool INV>
__device__ __forceinline__ void pass2_8(c32* lds, c32 wbase, int t) {
    // posA8(a'*512 + 32 b + low) = a'*528 + low + 33 b
    c32* base = lds + (t >> 5) * 528 + (t & 31);
    c32 x[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) x[b] = lds_ld(base + 33 * b);
    c32 w = wbase;
    SSK_OPAQUE2(w);
    if (INV) twiddle16<true>(x, w);
    fft16<INV>(x);
    if (!INV) twiddle16<false>(x, w);
#pragma unroll
    for (int b = 0; b < 16; ++b) lds_st(base + 33 * b, x[b]);
}

// ---- pass 3: thread = d*256 + ab; twiddle exp(-+2 pi i d c'/32), d in {0, 1} (wave-uniform) -----------------------
// exp(-2 pi i c'/32) = exp(-2 pi i 2c'/64): the D = 2 literal set of the 16384-point core
__device__ __forceinline__ void pass3_fwd8(c32* lds, int t) {
    const int d = t >> 8, ab = t & 255;
    const c32* src = lds + 33 * ab + d;           // posA8(ab*32 + 2c + d) = 33 ab + d + 2c
    c32* dst = lds + 4352 * d + 17 * ab;          // posB(d, ab, c')
    c32 x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = lds_ld(src + 2 * c);
    fft16<false>(x);
    const int du = __builtin_amdgcn_readfirstlane(d);
    if (du == 0) {                                 // each branch carries the rest of the pass (see ssk::pass3_fwd)
        lds_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) lds_st(dst + c, x[c]);
    } else {
        ssk::twiddle16_const<false, 2>(x);
        lds_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) lds_st(dst + c, x[c]);
    }
}
__device__ __forceinline__ void pass3_inv8(c32* lds, int t) {
    const int d = t >> 8, ab = t & 255;
    const c32* src = lds + 4352 * d + 17 * ab;
    c32* dst = lds + 33 * ab + d;
    c32 x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = lds_ld(src + c);
    const int du = __builtin_amdgcn_readfirstlane(d);
    if (du == 0) {
        fft16<true>(x);
        lds_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) lds_st(dst + 2 * c, x[c]);
    } else {
        ssk::twiddle16_const<true, 2>(x);
        fft16<true>(x);
        lds_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) lds_st(dst + 2 * c, x[c]);
    }
}

// ---- items: radix 2 over d fused with the Hermitian split ----------------------------------------------------------
// item q in [0, 2048): groups gA = item_gA(q), gB = 4096 - gA (item 0: groups 0 and 2048, self-paired).
// Forward: v[0] = X2[gA], v[1] = X2[gA + 4096], v[2] = X2[gB], v[3] = X2[gB + 4096]  (X2 = 2 * rFFT_16384 bins; item 0:
// v[0] = (X2[0], X2[8192]) both real).  Bin pairs (k, 8192 - k): (gA, gB + 4096) and (gA + 4096, gB).
__device__ __forceinline__ void item_load_fwd8(const c32* lds, c32 wbase, int q, c32 (&v)[4]) {
    const int gA = item_gA(q);
    const int gB = (q == 0) ? 2048 : 4096 - gA;
    const c32* pa = lds + 17 * group_ab(gA) + group_c(gA);
    const c32* pb = lds + 17 * group_ab(gB) + group_c(gB);
    const c32 a0 = lds_ld(pa), a1 = lds_ld(pa + 4352), b0 = lds_ld(pb), b1 = lds_ld(pb + 4352);
    v[0] = cadd(a0, a1); v[1] = csub(a0, a1);
    v[2] = cadd(b0, b1); v[3] = csub(b0, b1);
    c32 wg = wbase;                               // exp(-2 pi i gA / 16384)
    SSK_OPAQUE2(wg);
    if (q != 0) {
        herm_fwd(v[0], v[3], wg);
        herm_fwd(v[1], v[2], mk2(wg.y, -wg.x));                       // * exp(-2 pi i 4096/16384) = -i
    } else {
        constexpr float H = 0.70710678118654752f;
        const c32 v0 = v[0];
        v[0] = mk2(2.f * (v0.x + v0.y), 2.f * (v0.x - v0.y));          // X2[0], X2[8192]
        c32 dup = v[1]; herm_fwd(v[1], dup, mk2(0.f, -1.f));           // k = 4096 (self)
        herm_fwd(v[2], v[3], mk2(H, -H));                              // k = 2048 <-> 6144
    }
}
__device__ __forceinline__ void item_store_inv8(c32* lds, c32 wbase, int q, c32 (&y)[4]) {
    const int gA = item_gA(q);
    const int gB = (q == 0) ? 2048 : 4096 - gA;
    c32* pa = lds + 17 * group_ab(gA) + group_c(gA);
    c32* pb = lds + 17 * group_ab(gB) + group_c(gB);
    c32 wg = wbase;
    SSK_OPAQUE2(wg);
    if (q != 0) {
        herm_inv(y[0], y[3], wg);
        herm_inv(y[1], y[2], mk2(wg.y, -wg.x));
    } else {
        constexpr float H = 0.70710678118654752f;
        const c32 y0 = y[0];                                            // (Y2[0], Y2[8192])
        y[0] = mk2(y0.x + y0.y, y0.x - y0.y);
        c32 dup = y[1]; herm_inv(y[1], dup, mk2(0.f, -1.f));
        herm_inv(y[2], y[3], mk2(H, -H));
    }
    lds_st(pa, cadd(y[0], y[1])); lds_st(pa + 4352, csub(y[0], y[1]));
    lds_st(pb, cadd(y[2], y[3])); lds_st(pb + 4352, csub(y[2], y[3]));
}

// forward chain after pass 1 up to "layout B holds the radix-2 groups"
__device__ __forceinline__ void fwd_passes8(c32* lds, const ThreadTw8& tw, int t) {
    lds_barrier();
    pass2_8<false>(lds, tw.p2, t);
    lds_barrier();
    pass3_fwd8(lds, t);
    lds_barrier();
}
// inverse chain from "layout B holds the merged product spectrum" to the 16 packed pairs t + 512 a in registers
__device__ __forceinline__ void inv_passes8(c32* lds, const ThreadTw8& tw, int t, c32 (&x)[16]) {
    lds_barrier();
    pass3_inv8(lds, t);
    lds_barrier();
    pass2_8<true>(lds, tw.p2, t);
    lds_barrier();
    pass1_inv8(lds, tw.p1, t, x);
}

}  // namespace ssk8

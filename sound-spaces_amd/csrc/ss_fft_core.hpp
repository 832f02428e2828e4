// ss_fft_core.hpp — register/LDS FFT building blocks for gfx950 (MI355X).
//
// One workgroup of kT = 1024 threads (16 wave64) owns one 16384-point complex
// FFT held in LDS (136 KiB of the CU's 160 KiB).  16384 = 16*16*16*4: three
// radix-16 passes and one radix-4 pass, each thread holding 16 complex values
// in registers per pass; passes exchange data through LDS in two layouts that
// keep every ds_read_b64 / ds_write_b64 bank-conflict free:
//
//   layout A (between passes 1-2-3): posA(p) = p + (p >> 6)      (1 complex of
//            padding per 64; pass 3 reads it with lane stride 65)
//   layout B (between passes 3-4):   posB(d, ab, c) = d*4352 + ab*17 + c
//            where the logical position is p = ab*64 + c*4 + d
//
// A real 32768-sample block is transformed as a 16384-point complex FFT of
// the even/odd-packed signal followed by a Hermitian split that is done in
// registers: the last (radix-4) pass assigns to every thread two "items",
// each a pair of radix-4 groups (g, 4096-g) whose output bins are exactly
// each other's Hermitian partners (k <-> 16384-k), so the split, the spectral
// multiply and the inverse re-packing need no data exchange at all.
//
// This file is plain HIP.  tests/hostsim/ compiles it for the host against a
// fiber-based shim of threadIdx/__syncthreads (test infrastructure only) so
// the index algebra is checked on CPU; the product always runs the gfx950 build.
#pragma once

// Stops the compiler from common-subexpression-ing the (cheap) twiddle power chains of a forward pass
// with those of the matching inverse pass: keeping ~30 values alive across the whole kernel costs
// scratch spills at the 128-VGPR budget of a 1024-thread workgroup.  (No-op in the host-side test build.)
#if defined(__HIP_DEVICE_COMPILE__)
#define SSK_OPAQUE2(v) asm volatile("" : "+v"((v).x), "+v"((v).y))
#define SSK_OPAQUE1(v) asm volatile("" : "+v"(v))
#else
#define SSK_OPAQUE2(v) (void)(v)
#define SSK_OPAQUE1(v) (void)(v)
#endif

namespace ssk {

constexpr int kM = 16384;            // complex points of one block FFT
constexpr int kB = 16384;            // real samples of one partition block (FFT covers 2*kB)
constexpr int kT = 1024;             // threads per workgroup
constexpr int kLdsComplex = 4 * 4352;  // 17408 complex = 139264 B (layout B; layout A needs 16640)
constexpr int kSpecComplex = 16384;  // complex values of one stored block spectrum (kernel order)
constexpr int kTwM = 1024;           // entries of twM:  exp(-2*pi*i*t/16384), t < 1024
constexpr int kTwItem = 2048;        // entries of twItem: exp(-2*pi*i*gA(q)/32768), q < 2048

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// a * (INV ? conj(b) : b)
template <bool INV>
__device__ __forceinline__ float2 cmul_dir(float2 a, float2 b) { return INV ? cmulc(a, b) : cmul(a, b); }

// 4-point DFT in place, natural-order outputs.  INV = conjugate kernel (no scaling).
template <bool INV>
__device__ __forceinline__ void bfly4(float2& a, float2& b, float2& c, float2& d) {
    const float2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    c = csub(t0, t2);
    if (!INV) {
        b = make_float2(t1.x + t3.y, t1.y - t3.x);   // t1 - i*t3
        d = make_float2(t1.x - t3.y, t1.y + t3.x);   // t1 + i*t3
    } else {
        b = make_float2(t1.x - t3.y, t1.y + t3.x);
        d = make_float2(t1.x + t3.y, t1.y - t3.x);
    }
}

// x *= exp(-+2*pi*i*m/16) for the seven exponents a 4x4 Cooley-Tukey step needs.
template <bool INV, int MEXP>
__device__ __forceinline__ float2 tw16(float2 a) {
    constexpr float C = 0.92387953251128674f;   // cos(pi/8)
    constexpr float S = 0.38268343236508977f;   // sin(pi/8)
    constexpr float H = 0.70710678118654752f;   // sqrt(1/2)
    // forward twiddle value (re, im); inverse uses the conjugate
    if (MEXP == 0) return a;
    if (MEXP == 4) return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);          // -+i
    if (MEXP == 2) return INV ? make_float2(H * (a.x - a.y), H * (a.x + a.y))
                              : make_float2(H * (a.x + a.y), H * (a.y - a.x));            // (H, -+H)
    if (MEXP == 6) return INV ? make_float2(-H * (a.x + a.y), H * (a.x - a.y))
                              : make_float2(H * (a.y - a.x), -H * (a.x + a.y));           // (-H, -+H)
    float2 w = make_float2(1.f, 0.f);
    if (MEXP == 1) w = make_float2(C, -S);
    if (MEXP == 3) w = make_float2(S, -C);
    if (MEXP == 9) w = make_float2(-C, S);
    return cmul_dir<INV>(a, w);
}

// 16-point DFT of x[0..15]; result in NATURAL order in x (X[r] = sum_j x[j] W^{jr}).
template <bool INV>
__device__ __forceinline__ void fft16(float2 (&x)[16]) {
    // j = 4*j1 + j0, r = r1 + 4*r0
#pragma unroll
    for (int j0 = 0; j0 < 4; ++j0) bfly4<INV>(x[j0], x[j0 + 4], x[j0 + 8], x[j0 + 12]);
    // x[j0 + 4*r1] now holds A[j0][r1]; twiddle by W16^{j0*r1}
    x[5] = tw16<INV, 1>(x[5]);   x[9] = tw16<INV, 2>(x[9]);    x[13] = tw16<INV, 3>(x[13]);
    x[6] = tw16<INV, 2>(x[6]);   x[10] = tw16<INV, 4>(x[10]);  x[14] = tw16<INV, 6>(x[14]);
    x[7] = tw16<INV, 3>(x[7]);   x[11] = tw16<INV, 6>(x[11]);  x[15] = tw16<INV, 9>(x[15]);
#pragma unroll
    for (int r1 = 0; r1 < 4; ++r1) bfly4<INV>(x[4 * r1], x[4 * r1 + 1], x[4 * r1 + 2], x[4 * r1 + 3]);
    // x[4*r1 + r0] = X[r1 + 4*r0]  -> transpose the 4x4 register tile
    float2 t;
    t = x[1];  x[1] = x[4];   x[4] = t;
    t = x[2];  x[2] = x[8];   x[8] = t;
    t = x[3];  x[3] = x[12];  x[12] = t;
    t = x[6];  x[6] = x[9];   x[9] = t;
    t = x[7];  x[7] = x[13];  x[13] = t;
    t = x[11]; x[11] = x[14]; x[14] = t;
}

// x[r] *= w^r (INV: conj(w)^r), r = 1..15, powers built by a depth<=4 product chain.
template <bool INV>
__device__ __forceinline__ void twiddle16(float2 (&x)[16], float2 w) {
    if (INV) w = cconj(w);
    const float2 w2 = cmul(w, w), w3 = cmul(w2, w), w4 = cmul(w2, w2);
    x[1] = cmul(x[1], w);  x[2] = cmul(x[2], w2);  x[3] = cmul(x[3], w3);  x[4] = cmul(x[4], w4);
    const float2 w5 = cmul(w4, w), w6 = cmul(w3, w3), w7 = cmul(w4, w3), w8 = cmul(w4, w4);
    x[5] = cmul(x[5], w5);  x[6] = cmul(x[6], w6);  x[7] = cmul(x[7], w7);  x[8] = cmul(x[8], w8);
    x[9] = cmul(x[9], cmul(w8, w));    x[10] = cmul(x[10], cmul(w5, w5));
    x[11] = cmul(x[11], cmul(w8, w3)); x[12] = cmul(x[12], cmul(w6, w6));
    x[13] = cmul(x[13], cmul(w8, w5)); x[14] = cmul(x[14], cmul(w7, w7));
    x[15] = cmul(x[15], cmul(w8, w7));
}

// ---- LDS layouts ------------------------------------------------------------
__device__ __forceinline__ int posA(int p) { return p + (p >> 6); }
__device__ __forceinline__ int posB(int d, int ab, int c) { return d * 4352 + ab * 17 + c; }

// radix-4 group g in [0,4096): bins g + 4096*d'.  Digits g = a' + 16 b' + 256 c'.
__device__ __forceinline__ int group_ab(int g) { return ((g & 15) << 4) | ((g >> 4) & 15); }
__device__ __forceinline__ int group_c(int g) { return g >> 8; }

// Item q in [0,2048) -> first group gA; the second group is gB = 4096 - gA
// (item 0 is the special one: groups 0 and 2048, each self-paired).
// Lanes vary c' fastest so that layout-B reads are contiguous.
__device__ __forceinline__ int item_gA(int q) {
    const int c = q & 15, lo = q >> 4;
    if (lo != 0) return lo + 256 * c;
    if (c < 8) return 256 * c;               // c == 0 -> 0 (special)
    return 128 + 256 * (c - 8);
}

// ---- forward passes 2,3 and inverse passes 3',2' (LDS <-> LDS) ---------------
// All take the workgroup's LDS buffer and the twM table (global memory).

// pass 2 (forward, in place, layout A): a' = t>>6, low2 = t&63
template <bool INV>
__device__ __forceinline__ void pass2(float2* lds, const float2* __restrict__ twM, int t) {
    // posA(a'*1024 + b*64 + low2) = a'*1040 + low2 + 65*b : one base register + immediate offsets
    float2* base = lds + (t >> 6) * 1040 + (t & 63);
    float2 x[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) x[b] = base[65 * b];
    float2 w = twM[16 * (t & 63)];
    SSK_OPAQUE2(w);
    if (INV) twiddle16<true>(x, w);
    fft16<INV>(x);
    if (!INV) twiddle16<false>(x, w);
#pragma unroll
    for (int b = 0; b < 16; ++b) base[65 * b] = x[b];
}

// pass 3 forward: read layout A, write layout B.  thread = d*256 + ab.
__device__ __forceinline__ void pass3_fwd(float2* lds, const float2* __restrict__ twM, int t) {
    const int d = t >> 8, ab = t & 255;
    const float2* src = lds + 65 * ab + d;           // posA(ab*64 + 4c + d) = 65*ab + d + 4c
    float2* dst = lds + 4352 * d + 17 * ab;          // posB(d, ab, c)
    float2 x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = src[4 * c];
    fft16<false>(x);
    float2 w = twM[256 * d];
    SSK_OPAQUE2(w);
    twiddle16<false>(x, w);
    __syncthreads();                       // every layout-A read done before layout-B writes
#pragma unroll
    for (int c = 0; c < 16; ++c) dst[c] = x[c];
}

// pass 3 inverse: read layout B, write layout A.
__device__ __forceinline__ void pass3_inv(float2* lds, const float2* __restrict__ twM, int t) {
    const int d = t >> 8, ab = t & 255;
    const float2* src = lds + 4352 * d + 17 * ab;
    float2* dst = lds + 65 * ab + d;
    float2 x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = src[c];
    float2 w = twM[256 * d];
    SSK_OPAQUE2(w);
    twiddle16<true>(x, w);
    fft16<true>(x);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) dst[4 * c] = x[c];
}

// ---- Hermitian split / merge on one bin pair (k, 16384-k) ---------------------
// Forward:  (Vk, Vp) -> (X2[k], X2[16384-k]) = 2 * rFFT_32768 bins,  wk = exp(-2 pi i k / 32768).
__device__ __forceinline__ void herm_fwd(float2& vk, float2& vp, float2 wk) {
    const float2 P = make_float2(vk.x + vp.x, vk.y - vp.y);     // Vk + conj(Vp)
    const float2 Q = make_float2(vk.x - vp.x, vk.y + vp.y);     // Vk - conj(Vp)
    const float2 wq = cmul(wk, Q);
    const float2 R = make_float2(wq.y, -wq.x);                  // -i * wk * Q
    vk = cadd(P, R);
    vp = make_float2(P.x - R.x, -(P.y - R.y));                  // conj(P - R)
}
// Inverse:  (Yk, Yp) -> (V'2[k], V'2[16384-k]),  V' = packed spectrum of the real output.
__device__ __forceinline__ void herm_inv(float2& yk, float2& yp, float2 wk) {
    const float2 P = make_float2(yk.x + yp.x, yk.y - yp.y);
    const float2 Q = make_float2(yk.x - yp.x, yk.y + yp.y);
    const float2 wq = cmulc(Q, wk);                             // conj(wk) * Q
    const float2 R = make_float2(-wq.y, wq.x);                  // i * conj(wk) * Q
    yk = cadd(P, R);
    yp = make_float2(P.x - R.x, -(P.y - R.y));
}

// exp(-2 pi i d / 8), d = 0..3
__device__ __forceinline__ float2 w8(int d) {
    constexpr float H = 0.70710678118654752f;
    return d == 0 ? make_float2(1.f, 0.f) : d == 1 ? make_float2(H, -H)
         : d == 2 ? make_float2(0.f, -1.f) : make_float2(-H, -H);
}

// Pass 4 forward on one item: read the two radix-4 groups from layout B, finish the
// 16384-point FFT, and turn the 8 bins into 2*rFFT_32768 bins:
//   v[j]   (j<4)  = X2[gA + 4096 j]
//   v[4+j] (j<4)  = X2[gB + 4096 j]         (item 0: v[0] = (X2[0], X2[16384]) both real)
__device__ __forceinline__ void item_load_fwd(const float2* lds, const float2* __restrict__ twItem,
                                              int q, float2 (&v)[8]) {
    const int gA = item_gA(q);
    const int gB = (q == 0) ? 2048 : 4096 - gA;
    const float2* pa = lds + 17 * group_ab(gA) + group_c(gA);
    const float2* pb = lds + 17 * group_ab(gB) + group_c(gB);
#pragma unroll
    for (int d = 0; d < 4; ++d) { v[d] = pa[4352 * d]; v[4 + d] = pb[4352 * d]; }
    bfly4<false>(v[0], v[1], v[2], v[3]);
    bfly4<false>(v[4], v[5], v[6], v[7]);
    float2 wg = twItem[q];                 // exp(-2 pi i gA / 32768)
    SSK_OPAQUE2(wg);
    if (q != 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d) herm_fwd(v[d], v[7 - d], cmul(wg, w8(d)));
    } else {
        constexpr float C = 0.92387953251128674f, S = 0.38268343236508977f;
        const float2 v0 = v[0];
        v[0] = make_float2(2.f * (v0.x + v0.y), 2.f * (v0.x - v0.y));      // X2[0], X2[16384]
        herm_fwd(v[1], v[3], w8(1));                                       // k = 4096
        float2 dup = v[2]; herm_fwd(v[2], dup, w8(2));                     // k = 8192 (self)
        herm_fwd(v[4], v[7], make_float2(C, -S));                          // k = 2048
        herm_fwd(v[5], v[6], make_float2(S, -C));                          // k = 6144
    }
}

// Inverse of the above: y[] holds Y2 bins in the same slot order; produce packed
// spectrum V'2, run the inverse radix-4 and write both groups back to layout B.
__device__ __forceinline__ void item_store_inv(float2* lds, const float2* __restrict__ twItem,
                                               int q, float2 (&y)[8]) {
    const int gA = item_gA(q);
    const int gB = (q == 0) ? 2048 : 4096 - gA;
    float2* pa = lds + 17 * group_ab(gA) + group_c(gA);
    float2* pb = lds + 17 * group_ab(gB) + group_c(gB);
    float2 wg = twItem[q];
    SSK_OPAQUE2(wg);
    if (q != 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d) herm_inv(y[d], y[7 - d], cmul(wg, w8(d)));
    } else {
        constexpr float C = 0.92387953251128674f, S = 0.38268343236508977f;
        const float2 y0 = y[0];                                            // (Y2[0], Y2[16384])
        y[0] = make_float2(y0.x + y0.y, y0.x - y0.y);
        herm_inv(y[1], y[3], w8(1));
        float2 dup = y[2]; herm_inv(y[2], dup, w8(2));
        herm_inv(y[4], y[7], make_float2(C, -S));
        herm_inv(y[5], y[6], make_float2(S, -C));
    }
    bfly4<true>(y[0], y[1], y[2], y[3]);
    bfly4<true>(y[4], y[5], y[6], y[7]);
#pragma unroll
    for (int d = 0; d < 4; ++d) { pa[4352 * d] = y[d]; pb[4352 * d] = y[4 + d]; }
}

}  // namespace ssk

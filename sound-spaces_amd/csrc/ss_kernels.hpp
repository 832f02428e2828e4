// ss_kernels.hpp — the three gfx950 kernels of the SoundSpaces audio-observation path.
//
//   k_source_windows : source clip window  -> block spectrum S' (kernel order), cached per (sound, t0)
//   k_conv           : RIR bank entry      -> audiogoal [N,2,out_len]  (reference: fftconvolve x2 + slice,
//                                             soundspaces/simulator.py:629-647, 649-664) and, fused,
//                                             the spectrogram [N,65,T4,2] (soundspaces/tasks/nav.py:86-100)
//   k_spectrogram    : audiogoal           -> spectrogram (stand-alone; 44.1 kHz and AudioGoal-only callers)
//
// Convolution model (all reference windowing branches are this one formula, see oracle/ss_oracle.py):
//   out[c,t] = sum_k h[c,k] * x[t0 + t - k],   x[n<0] = 0
// computed as uniformly-partitioned overlap-save with block kB = 16384 (FFT of 2*kB real samples =
// one 16384-point complex FFT in LDS):  Y_j = sum_i H_i * S_{j-i},  S_m = rFFT(x[t0+(m-1)kB : t0+(m+1)kB]),
// out block j = last kB samples of irFFT(Y_j).  At 16 kHz with a <= 1.024 s RIR this is one forward and
// one inverse FFT per (unit, ear); the source spectra S_m are precomputed per (sound, t0) and shared.
#pragma once
#include "ss_fft_core.hpp"

namespace ssk {

// wave-uniform words through the scalar cache (s_load, counted by lgkmcnt - independent of the vector-memory queue)
__device__ __forceinline__ int uniform_load(const int* ptr) {      // wave-uniform address -> scalar cache
#if defined(__HIP_DEVICE_COMPILE__)
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    return v;
#else
    return *ptr;
#endif
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 uniform_load4(const int* ptr) {   // 4 consecutive words, wave-uniform address
#if defined(__HIP_DEVICE_COMPILE__)
    int a, b, c, d;                     // four dword loads, ONE wait (no alignment requirement beyond 4 bytes)
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x4\n\ts_load_dword %2, %4, 0x8\n\t"
                 "s_load_dword %3, %4, 0xc\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(ptr) : "memory");
    return i32x4{a, b, c, d};
#else
    return i32x4{ptr[0], ptr[1], ptr[2], ptr[3]};
#endif
}

// both terms of a unit descriptor (8 words) in ONE scalar round trip
__device__ __forceinline__ void uniform_load8(const int* ptr, i32x4& lo, i32x4& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    int a, b, c, d, e, f, g, h;
    asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x4\n\ts_load_dword %2, %8, 0x8\n\t"
                 "s_load_dword %3, %8, 0xc\n\ts_load_dword %4, %8, 0x10\n\ts_load_dword %5, %8, 0x14\n\t"
                 "s_load_dword %6, %8, 0x18\n\ts_load_dword %7, %8, 0x1c\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d), "=&s"(e), "=&s"(f), "=&s"(g), "=&s"(h) : "s"(ptr) : "memory");
    lo = i32x4{a, b, c, d};
    hi = i32x4{e, f, g, h};
#else
    lo = i32x4{ptr[0], ptr[1], ptr[2], ptr[3]};
    hi = i32x4{ptr[4], ptr[5], ptr[6], ptr[7]};
#endif
}

__device__ __forceinline__ f32x4 mk4(c32 a, c32 b) { f32x4 r; r.xy = a; r.zw = b; return r; }

// Streamed-once global accesses (a RIR row is read by ONE workgroup once per step, an audiogoal row written once):
// the nt policy keeps them from displacing the window spectra, which every step re-reads, from the XCD's 4 MiB L2
// (MI355X_MICROARCH price list "nt-weights": -18 % issue-to-landed for data one CU reads once).
template <class T>
__device__ __forceinline__ T ld_stream(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <class T>
__device__ __forceinline__ void st_stream(T* p, T v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// STFT geometry of SpectrogramSensor.compute_spectrogram (nav.py:88-93)
constexpr int kNfft = 512, kHop = 160, kPool = 4, kBins4 = 65;   // 257 bins -> 65 pooled rows
constexpr int kPrevPairs = 1208;          // XFADE: packed pairs of the cross-fade ramp kept per row (fade_len <= 2414: 48 kHz)
constexpr int kFrameStride = 272;         // transpose tiles (16 x 17 complex per frame); = 16 mod 32 so that the four
                                          // frames of a wave start 32 banks apart (ds_read/write_b64 rules)
constexpr int kNatStride = 288;           // natural-order spectra per frame; = 0 mod 32 (ds_read_b128 lane groups)
constexpr int kWaveScratch = 4 * kNatStride;     // complex per wave: 4 frames
constexpr int kPsStride = 112;            // floats per frame of pooled partial sums (>= 97, = 16 mod 64: bank-disjoint)
// Natural-order spectrum / tw512 table index: 2 complex (16 B) of padding after every 32.  The magnitude stage reads
// Z[4b..4b+1], lane = b, as ds_read_b128 at a 32-byte lane stride; unpadded, the 16 lanes of a b128 lane group
// ({0-3,12-15,20-27}, ... : MI355X_MICROARCH LDS table) hit 8 four-bank groups twice (measured SQ_LDS_BANK_CONFLICT
// = 16 % of the STFT phase's LDS cycles).  With this padding and kNatStride every b128 group covers all 64 banks once
// (tests/test_lds_banks.py enumerates it).
__host__ __device__ constexpr int posN(int k) { return k + 2 * (k >> 5); }
constexpr int kTw512Lds = posN(255) + 1;  // 270 complex

// Device-resident constant tables (built once per device by the host library, in double precision).
struct Tables {
    const c32* twM;      // [1024]  exp(-2 pi i t / 16384)
    const c32* twItem;   // [2048]  exp(-2 pi i gA(q) / 32768)
    const c32* tw512;    // [256]   exp(-2 pi i k / 512)
    const float*  win;      // [512]   hann(400, periodic) centred in 512
    const c32* twG;      // [512]   exp(-2 pi i q / 32768)             (512-thread core, ss_fft_core32.hpp)
    const c32* twP2;     // [512]   exp(-2 pi i c k2 / 512) at [c * 32 + k2]   (its pass-2 twiddles)
};

// ---------------------------------------------------------------------------------------------
// pass 1 forward (global -> LDS layout A).  LOADER(m) returns the packed sample pair (x[2m], x[2m+1]).
// HALF: packed samples m >= 8192 are known to be zero (an RIR block has <= kB real samples).
template <bool HALF, class LOADER>
__device__ __forceinline__ void pass1_fwd(c32* lds, c32 wbase, int t, LOADER load) {
    c32 x[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) x[a] = (HALF && a >= 8) ? mk2(0.f, 0.f) : load(t + 1024 * a);
    if (HALF) fft16_fwd_lo8(x); else fft16<false>(x);
    c32 w = wbase;
    SSK_OPAQUE2(w);
    twiddle16<false>(x, w);
    c32* base = lds + t + (t >> 6);             // posA(t + 1024*a) = t + (t>>6) + 1040*a
#pragma unroll
    for (int a = 0; a < 16; ++a) lds_st(base + 1040 * a, x[a]);
}

// pass 1 inverse: LDS layout A -> registers; only the upper half (packed samples 8192..16383, i.e. the
// alias-free last kB real samples of the circular convolution) is produced: y[a-8] <-> packed m = t+1024*(a-8).
__device__ __forceinline__ void pass1_inv(const c32* lds, c32 wbase, int t, c32 (&y)[8]) {
    c32 x[16];
    const c32* base = lds + t + (t >> 6);
#pragma unroll
    for (int a = 0; a < 16; ++a) x[a] = lds_ld(base + 1040 * a);
    c32 w = wbase;
    SSK_OPAQUE2(w);
    twiddle16<true>(x, w);
    fft16<true>(x);
#pragma unroll
    for (int a = 0; a < 8; ++a) y[a] = x[8 + a];
}

// forward chain after pass 1 up to "layout B holds the radix-4 groups" (items are then read one at a time)
__device__ __forceinline__ void fwd_passes(c32* lds, const ThreadTw& tw, int t) {
    lds_barrier();
    pass2<false>(lds, tw.p2, t);
    lds_barrier();
    pass3_fwd(lds, t);
    lds_barrier();
}

// inverse chain from "items hold Y2 bins" to the last kB real samples in registers.
// LDS must not be in use by other threads' pending reads (caller syncs before).
__device__ __forceinline__ void items_to_time(c32* lds, const ThreadTw& tw, int t, c32 (&acc)[2][8], c32 (&y)[8]) {
    item_store_inv(lds, tw.i0, t, acc[0]);
    item_store_inv(lds, tw.i1, t + 1024, acc[1]);
    lds_barrier();
    pass3_inv(lds, t);
    lds_barrier();
    pass2<true>(lds, tw.p2, t);
    lds_barrier();
    pass1_inv(lds, tw.p1, t, y);
}

// ---------------------------------------------------------------------------------------------
// k_source_windows: one workgroup per window.  desc[w] = {src_offset, src_len, start, wrap}.
// Window sample n (0 <= n < 32768) is x[start + n], 0 outside [0, src_len) unless wrap (then indices
// >= src_len continue at the beginning of the clip: continuous_simulator.py:441-445).
// Output: spec[w] = 8192 f32x4 = S'/(4*16384) in kernel order: thread t, item s, slot pair h (slots 2h, 2h+1)
// at f32x4 index (s*4+h)*1024 + t, so that every consumer load is one coalesced 16-byte access per lane.
struct SrcParams {
    const float* src;
    const int* desc;      // [W][desc_stride]; desc_stride == 5: word 4 = output slot (scattered into a pool), else slot = w
    f32x4* spec;         // [slots][8192]
    Tables tb;
    int desc_stride;      // 4 or 5
    float scale;          // factor applied to 2*rFFT (window spectra: 1/(8*16384), see k_source_windows)
};

constexpr float kWindowScale = 1.0f / (8.0f * 16384.0f);   // (2X -> X) * 1/(4M): inverse packing 2x2, 1/M of the IFFT

__device__ __forceinline__ float src_sample(const float* __restrict__ x, int len, int s, int wrap) {
    if (s < 0) return 0.f;
    if (s >= len) {
        if (!wrap) return 0.f;
        s -= len;
        if (s >= len) return 0.f;
    }
    return x[s];
}

__global__ __launch_bounds__(1024) void k_source_windows(SrcParams p) {
    __shared__ c32 lds[kLdsComplex];
    const int t = threadIdx.x, w = blockIdx.x;
    const int* d = p.desc + p.desc_stride * w;
    const float* x = p.src + __builtin_amdgcn_readfirstlane(d[0]);
    const int len = __builtin_amdgcn_readfirstlane(d[1]), start = __builtin_amdgcn_readfirstlane(d[2]);
    const int wrap = __builtin_amdgcn_readfirstlane(d[3]);
    const int slot = p.desc_stride > 4 ? __builtin_amdgcn_readfirstlane(d[4]) : w;
    const ThreadTw tw = load_thread_tw(p.tb.twM, p.tb.twItem, t);
    pass1_fwd<false>(lds, tw.p1, t, [&](int m) {
        return mk2(src_sample(x, len, start + 2 * m, wrap), src_sample(x, len, start + 2 * m + 1, wrap));
    });
    fwd_passes(lds, tw, t);
    const float scale = p.scale;                         // window spectra: (2X -> X) * 1/(4M) = 1/(8*16384)
    f32x4* o = p.spec + (size_t)slot * (kSpecComplex / 2) + t;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        c32 v[8];
        item_load_fwd(lds, s ? tw.i1 : tw.i0, t + 1024 * s, v);
#pragma unroll
        for (int h = 0; h < 4; ++h)
            o[(s * 4 + h) * 1024] = mk4(v[2 * h] * scale, v[2 * h + 1] * scale);
    }
}

// ---------------------------------------------------------------------------------------------
// STFT -> |.| -> 4x4 mean-pool -> log1p for one 4-frame time block per wave.
// Frame tf covers y[160*tf-256 .. 160*tf+255] of the audiogoal row y (librosa centre padding: reflect (pad_mode 0) or
// zeros (pad_mode 1)); every kernel materialises the padding once, in LDS, before the frames are read.
// phase A: windowed frame samples -> registers (packed even/odd), lane = f*16 + q, for a row that sits in LDS WITH its
// centre padding materialised: every live frame is 16 aligned 8-byte reads of the row and 16 of the window table
__device__ __forceinline__ void stft_load_padded(const float* padded, int tf, int n_frames, int q, const float* win,
                                                 c32 (&x)[16]) {
    if (tf >= n_frames) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = mk2(0.f, 0.f);
    } else {
        const c32* y2 = reinterpret_cast<const c32*>(padded + kHop * tf) + q;
        const c32* w2 = reinterpret_cast<const c32*>(win) + q;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const c32 s = lds_ld(y2 + 16 * j), w = lds_ld(w2 + 16 * j);
            x[j] = mk2(w.x * s.x, w.y * s.y);
        }
    }
}

// the same with the lane's 16 window pairs already in registers (they depend on q alone: a wave that loads two blocks reads
// them once - LDS reads are what the hand-off of the fused kernels waits for: 16 fewer of 64 per lane)
__device__ __forceinline__ void stft_window_pairs(const float* win, int q, c32 (&w)[16]) {
    const c32* w2 = reinterpret_cast<const c32*>(win) + q;
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = lds_ld(w2 + 16 * j);
}
__device__ __forceinline__ void stft_load_padded_w(const float* padded, int tf, int n_frames, int q, const c32 (&w)[16],
                                                   c32 (&x)[16]) {
    if (tf >= n_frames) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = mk2(0.f, 0.f);
    } else {
        const c32* y2 = reinterpret_cast<const c32*>(padded + kHop * tf) + q;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const c32 s = lds_ld(y2 + 16 * j);
            x[j] = mk2(w[j].x * s.x, w[j].y * s.y);
        }
    }
}

// phase B: everything after the load, for ONE wave: sc = this wave's private scratch (kWaveScratch complex),
// so all synchronisation is wave-scope (no workgroup barrier).  Pooled log1p values go out through STORE(b, value).
// wq = exp(-2 pi i q / 256) (= twM[64 q], loaded once by the caller); tw512 = LDS copy of the table exp(-2 pi i k / 512), entry k at posN(k).
template <class STORE>
__device__ __forceinline__ void stft_block(c32* sc, int lane, c32 wq, const c32* tw512, c32 (&x)[16], STORE store) {
    const int f = lane >> 4, q = lane & 15;
    c32* fr = sc + f * kFrameStride;        // transpose tile of this frame
    c32* fn = sc + f * kNatStride;          // natural-order spectrum of this frame (after the second pass)
    // 256-point FFT of the packed frame: pass 1 over j (stride 16), twiddle w256^(q r), transpose, pass 2 over q
    fft16<false>(x);
    SSK_OPAQUE2(wq);
    twiddle16<false>(x, wq);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_st(fr + r * 17 + q, x[r]);
    wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = lds_ld(fr + q * 17 + r);
    fft16<false>(x);                         // x[s] = Z[q + 16 s]
    wave_sync();
#pragma unroll
    for (int s = 0; s < 16; ++s) lds_st(fn + q + posN(16 * s), x[s]);   // posN(q + 16 s) = q + posN(16 s)
    wave_sync();
    // |rFFT_512| pooled over 4 bins.  Bins come in Hermitian pairs (k, 256-k) that share P, Q and w*Q, so a lane
    // takes the pooled rows b = q and q+16 (bins 4b..4b+3 < 128) together with their mirror bins 256-4b-e:
    //   D[b]    = sum_e |X[4b+e]|                      -> row b
    //   M0[b]   = |X[256-4b]|                           -> row 64-b
    //   M123[b] = sum_{e=1..3} |X[256-4b-e]|            -> row 63-b          (+ |X[128]| = |Z[128]| for row 32)
    float* ps = reinterpret_cast<float*>(sc);           // per frame: D[32] | M0[32] | M123[32] | X128 ; stride kPsStride
    float dsum[2], m0[2], m123[2];
    // Z[256-4b], the mirror of Z[4b], sits just above this lane's aligned mirror quad: it is the previous lane's
    // Z[252-4(b-1)], fetched by DPP (a direct LDS read at this lane stride would be 4-way bank conflicted)
    f32x4 k01[2], k23[2], p01[2], p23[2], w01[2], w23[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = q + 16 * i;
        const f32x4* zk4 = reinterpret_cast<const f32x4*>(fn + posN(4 * b));          // Z[4b .. 4b+3]
        const f32x4* zp4 = reinterpret_cast<const f32x4*>(fn + posN(252 - 4 * b));    // Z[252-4b .. 255-4b]
        const f32x4* w4 = reinterpret_cast<const f32x4*>(tw512 + posN(4 * b));
        k01[i] = zk4[0]; k23[i] = zk4[1]; p01[i] = zp4[0]; p23[i] = zp4[1]; w01[i] = w4[0]; w23[i] = w4[1];
    }
    const c32 prev0 = mk2(row_ror1(p01[0].x, lane), row_ror1(p01[0].y, lane));         // lane q-1 (15 for q = 0), i = 0
    const c32 prev1 = mk2(row_ror1(p01[1].x, lane), row_ror1(p01[1].y, lane));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // b = 0: Z[256] = Z[0];  b = 16 (q = 0, i = 1): Z[192] is lane 15's i = 0 quad start
        const c32 ptop = i == 0 ? (q == 0 ? k01[0].xy : prev0) : (q == 0 ? prev0 : prev1);
        const c32 zk[4] = {k01[i].xy, k01[i].zw, k23[i].xy, k23[i].zw};
        const c32 zp[4] = {ptop, p23[i].zw, p23[i].xy, p01[i].zw};
        const c32 ww[4] = {w01[i].xy, w01[i].zw, w23[i].xy, w23[i].zw};
        float d = 0.f, m = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const c32 P = add_conj(zk[e], zp[e]), Q = sub_conj(zk[e], zp[e]);
            const c32 wq = cmul(Q, ww[e]);
            // X = P - i wq = 2 X[k], Y = P + i wq = conj(2 X[256-k]), by component; m2 = (|X|^2, |Y|^2)
            const c32 m2 = mag2(xy_re(P, wq), xy_im(P, wq));
            d += fast_sqrt(m2.x);
            const float my = fast_sqrt(m2.y);
            if (e == 0) m0[i] = 0.5f * my; else m += my;
        }
        dsum[i] = 0.5f * d;
        m123[i] = 0.5f * m;
    }
    float x128 = 0.f;
    if (q == 0) { const c32 z = fn[posN(128)]; x128 = fast_sqrt(z.x * z.x + z.y * z.y); }
    wave_sync();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ps[f * kPsStride + q + 16 * i] = dsum[i];
        ps[f * kPsStride + 32 + q + 16 * i] = m0[i];
        ps[f * kPsStride + 64 + q + 16 * i] = m123[i];
    }
    if (q == 0) ps[f * kPsStride + 96] = x128;
    if (q == 1) ps[f * kPsStride + 97] = 0.f;            // the "no second term" slot of the gather below
    wave_sync();
    // Pooled row r = lane (0..63) is the sum over the 4 frames of ONE or TWO partial sums:
    //   r < 32: D[r]      r == 32: M123[31] + X128      32 < r < 64: M123[63-r] + M0[64-r]
    // as a branch-free gather through two per-lane offsets (the absent second term reads the zero slot; x + 0 is exact):
    // eight independent LDS reads and one wait.  (Written as a chain of if / else per frame it compiled to exec-masked
    // branches with an LDS read and an s_waitcnt lgkmcnt(0) in every arm: a dozen dependent LDS round trips per block.)
    {
        const int r = lane;
        const bool lo = r < 32, mid = r == 32;
        const int oa = lo ? r : mid ? 64 + 31 : 64 + 63 - r;
        const int ob = lo ? 97 : mid ? 96 : 32 + 64 - r;
        float a[4], b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { a[g] = ps[g * kPsStride + oa]; b[g] = ps[g * kPsStride + ob]; }
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) v += a[g] + b[g];
        store(r, fast_log1p(v * (1.0f / 16.0f)));
    }
    if (lane == 0) {                                     // row 64 = bin 256 alone: M0[0] of the 4 frames
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) v += ps[g * kPsStride + 32];
        store(64, fast_log1p(v * (1.0f / 16.0f)));
    }
}

struct SpecParams {
    const float* x;       // [N][2][len]
    float* out;           // [N][65][T4][2]
    Tables tb;
    int len, n_frames, t4, pad_mode;
    int gpw;              // groups (of 4 pooled time blocks) handled per workgroup
    int live;             // pooled blocks >= live are KNOWN to be zero (rows rendered up to n_valid < len: see live_blocks);
};                        // t4 when nothing is known about the rows

// pooled time blocks that can be non-zero when a row of `len` samples is zero from sample n_valid on: block b is exactly
// zero once all of its frames start behind the rendered samples (640 b - 256 >= n_valid) - provided the right centre
// padding mirrors zeros as well (n_valid <= len - 512); |STFT| of zeros is 0 and log1p(0) = 0
__host__ __device__ constexpr int live_blocks(int n_valid, int len, int t4) {
    return n_valid <= len - kNfft ? ((n_valid + kNfft / 2 + kHop * kPool - 1) / (kHop * kPool) < t4
                                         ? (n_valid + kNfft / 2 + kHop * kPool - 1) / (kHop * kPool) : t4)
                                  : t4;
}

// stand-alone spectrogram: 512 threads = 8 waves = {ear 0, ear 1} x 4 consecutive pooled time blocks of one unit.
// (Ablation of the previous one-wave-per-block layout, profiles/r1/NOTES.md: 58 of 200 us were the per-lane frame
// loads from global with their padding logic, 29 us the scattered 4-byte stores, 17 us all of the math.)
//   1. the 16 frames of the group span 2912 samples per ear: both segments are staged in LDS by coalesced loads, with
//      librosa's centre padding (reflect / zeros) resolved once per sample instead of once per frame lane;
//   2. every wave pulls its 4 frames from LDS into registers (always the branch-free path), then the staging area is
//      dead and is overlaid by the per-wave STFT scratch;
//   3. the 65 x (4 blocks x 2 ears) results are collected in LDS and written as contiguous 32-byte runs of the
//      channel-last output instead of 4-byte scatters.
constexpr int kSegFrames = 16, kSegLen = kHop * (kSegFrames - 1) + kNfft;     // 2912 samples
constexpr int kSegQuads = kSegLen / 4;                                         // 728 float4 per ear

// the (up to) 3 float4 of the two-ear segment of group g that thread t stages: quad e4 = t + 512 k.
// Interior groups (the segment lies inside the row; workgroup-uniform test) take a path WITHOUT per-lane branches:
// three unconditional 16-byte loads from clamped addresses and nothing that touches the loaded values.  With the per-lane
// "in range ? vector load : four padded scalar loads" form the compiler closed every quad with s_waitcnt vmcnt(0)
// before issuing the next (seen in the ISA): three serialised HBM round trips, also in the "prefetch" of the next
// group, which therefore ran in front of the math instead of under it.
__device__ __forceinline__ void spec_seg_load(const SpecParams& p, const float* row0, int g, int t, f32x4 (&r)[3]) {
    const int s0 = kHop * kSegFrames * g - kNfft / 2;
    const bool vec_ok = !(p.len & 3) && !(reinterpret_cast<size_t>(row0) & 15);
    if (vec_ok && s0 >= 0 && s0 + kSegLen <= p.len) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e4 = t + 512 * k;
            const int e = e4 < 2 * kSegQuads ? e4 : 0;      // clamp: the load is unconditional
            const int c = e >= kSegQuads, n = s0 + 4 * (e - c * kSegQuads);
            // no select on the loaded value (it would force the wait here): quads >= 2*kSegQuads are never stored
            r[k] = *reinterpret_cast<const f32x4*>(row0 + (size_t)c * p.len + n);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {                           // first / last group of a row, odd lengths, unaligned rows
        const int e4 = t + 512 * k;
        const int c = e4 >= kSegQuads, n = s0 + 4 * (e4 - c * kSegQuads);
        const float* row = row0 + (size_t)c * p.len;
        if (e4 >= 2 * kSegQuads) {
            r[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if (vec_ok && n >= 0 && n + 4 <= p.len) {
            r[k] = *reinterpret_cast<const f32x4*>(row + n);
        } else {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                   // librosa centre padding, resolved once per sample
                int i = n + u;
                if (p.pad_mode == 0) { i = i < 0 ? -i : i; i = i >= p.len ? 2 * (p.len - 1) - i : i; }
                const bool ok = i >= 0 && i < p.len;
                const float q = row[ok ? i : 0];
                v[u] = ok ? q : 0.f;
            }
            r[k] = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

__global__ __launch_bounds__(512) void k_spectrogram(SpecParams p) {
    alignas(16) __shared__ c32 sc[8 * kWaveScratch];        // 73728 B: staging (2 x 2912 floats), then 8 wave scratches
    __shared__ float res[kBins4 * 8];                       // [65][4 blocks][2 ears]
    __shared__ float s_win[kNfft];
    __shared__ c32 s_tw512[kTw512Lds];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int groups = (p.t4 + 3) >> 2, chunks = (groups + p.gpw - 1) / p.gpw;
    const int unit = blockIdx.x / chunks, g0 = (blockIdx.x % chunks) * p.gpw, g1_all = min(groups, g0 + p.gpw);
    // groups whose four pooled blocks are all known to be zero are written, not computed (SS2.0 steps: 0.25 s of a 1-s row)
    const int g1 = min(g1_all, (p.live + 3) >> 2);
    for (int g = max(g0, g1); g < g1_all; ++g) {
        if (t < kBins4 * 4) {
            const int b = t >> 2, c2 = t & 3;
            if (4 * g + c2 < p.t4)
                *reinterpret_cast<c32*>(p.out + ((size_t)unit * kBins4 + b) * p.t4 * 2 + 8 * g + 2 * c2) = mk2(0.f, 0.f);
        }
    }
    if (g0 >= g1) return;                                   // (workgroup-uniform, before any barrier)
    const int ch = wv >> 2, tbl = wv & 3;                   // this wave: ear, local pooled block
    const float* row0 = p.x + (size_t)unit * 2 * p.len;
    f32x4* seg4 = reinterpret_cast<f32x4*>(sc);             // seg[c][i] = padded y_c[160*16*g - 256 + i]
    const float* seg = reinterpret_cast<const float*>(sc);
    f32x4 r[3];
    spec_seg_load(p, row0, g0, t, r);
    s_win[t] = p.tb.win[t];
    if (t < 256) s_tw512[posN(t)] = p.tb.tw512[t];
    c32 wq = p.tb.twM[64 * (lane & 15)];
    // waited for HERE: if this load were still "pending" for the compiler inside the loop, its first use there would
    // get an s_waitcnt vmcnt(0), which (one in-order counter) also waits for the next segment's prefetch
    SSK_OPAQUE2(wq);
    for (int g = g0; g < g1; ++g) {
        // the barrier at the end of the previous round made the scratch (and res) dead
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (t + 512 * k < 2 * kSegQuads) seg4[t + 512 * k] = r[k];
        lds_barrier();
        const int tb4 = 4 * g + tbl, fl = 4 * tbl + (lane >> 4);
        const bool live = tb4 < p.t4 && kSegFrames * g + fl < p.n_frames;
        c32 x[16];
        {   // frame fl of the segment starts at seg[160*fl], padding already resolved: always the aligned path
            const c32* y2 = reinterpret_cast<const c32*>(seg + ch * kSegLen + kHop * fl) + (lane & 15);
            const c32* w2 = reinterpret_cast<const c32*>(s_win) + (lane & 15);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const c32 s = lds_ld(y2 + 16 * j), w = lds_ld(w2 + 16 * j);
                x[j] = live ? mk2(w.x * s.x, w.y * s.y) : mk2(0.f, 0.f);
            }
        }
        lds_barrier();                                      // staging area dead: scratch may overlay it
        if (g + 1 < g1) spec_seg_load(p, row0, g + 1, t, r);   // next segment in flight under this round's math
        if (tb4 < p.t4)
            stft_block(sc + wv * kWaveScratch, lane, wq, s_tw512, x, [&](int b, float v) { res[b * 8 + tbl * 2 + ch] = v; });
        lds_barrier();
        if (t < kBins4 * 4) {                               // 65 rows x 4 (block, ear-pair) float2 = 32-byte runs
            const int b = t >> 2, c2 = t & 3;
            if (4 * g + c2 < p.t4)
                *reinterpret_cast<c32*>(p.out + ((size_t)unit * kBins4 + b) * p.t4 * 2 + 8 * g + 2 * c2) =
                    *reinterpret_cast<const c32*>(res + 8 * b + 2 * c2);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_logmel (EXTENSION, not in the reference: SURVEY 8(f) rank 4 / BASELINE north_star "log-mel"):
//   out[n][j][tf][c] = log( sum_k W[j][k] |STFT_c[k][tf]|^2 + eps ),  same framing / window / padding as k_spectrogram,
// no pooling.  W comes as a band-sparse table: band j covers bins start[j] .. start[j]+max_len-1 with weights
// w[j][0..max_len) (zero padded; triangular mel filters are contiguous in k).
// Same workgroup shape and staging as k_spectrogram (8 waves = 2 ears x 4 blocks of 4 frames, padded segments staged in
// LDS, next group prefetched).  After the two radix-16 passes the power spectrum of a wave's 4 frames goes to LDS
// (kPowStride floats per frame, zero tail so that every band can run the same max_len-step loop), lane (f, q)
// accumulates bands q, q+16, ... of frame f, and the [n_mels][16 frames][2 ears] results of a group leave as 128-byte
// rows.  The filter bank is a sparse GEMV of ~2 non-zeros per bin: plain VALU FMAs fed from LDS (an MFMA tile would be
// >90 % zeros).
struct MelParams {
    const float* x;        // [N][2][len]
    float* out;            // [N][n_mels][n_frames][2]
    Tables tb;
    const int* start;      // [n_mels]
    const float* w;        // [n_mels][max_len]
    int len, n_frames, pad_mode, gpw;
    int n_mels, max_len;
    float eps;
};
constexpr int kPowStride = 320, kMelMaxLen = 64, kMelMaxBands = 128;   // start <= 256 (a multiple of 4): start + max_len <= 320
constexpr int kMelTableFloats = 4096;   // LDS copy of w (n_mels * max_len <= 4096)

// first half of stft_block: 256-point FFT of the packed frame, then the power spectrum |X[k]|^2, k = 0..256, of this
// lane's bins into pw[] (LDS, frame-private) -- see stft_block for the pairing of bins
__device__ __forceinline__ void stft_power(c32* sc, int lane, c32 wq, const c32* tw512, c32 (&x)[16]) {
    const int f = lane >> 4, q = lane & 15;
    c32* fr = sc + f * kFrameStride;
    c32* fn = sc + f * kNatStride;
    fft16<false>(x);
    SSK_OPAQUE2(wq);
    twiddle16<false>(x, wq);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_st(fr + r * 17 + q, x[r]);
    wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = lds_ld(fr + q * 17 + r);
    fft16<false>(x);
    wave_sync();
#pragma unroll
    for (int s = 0; s < 16; ++s) fn[q + posN(16 * s)] = x[s];
    wave_sync();
    f32x4 k01[2], k23[2], p01[2], p23[2], w01[2], w23[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = q + 16 * i;
        const f32x4* zk4 = reinterpret_cast<const f32x4*>(fn + posN(4 * b));
        const f32x4* zp4 = reinterpret_cast<const f32x4*>(fn + posN(252 - 4 * b));
        const f32x4* w4 = reinterpret_cast<const f32x4*>(tw512 + posN(4 * b));
        k01[i] = zk4[0]; k23[i] = zk4[1]; p01[i] = zp4[0]; p23[i] = zp4[1]; w01[i] = w4[0]; w23[i] = w4[1];
    }
    const c32 prev0 = mk2(row_ror1(p01[0].x, lane), row_ror1(p01[0].y, lane));
    const c32 prev1 = mk2(row_ror1(p01[1].x, lane), row_ror1(p01[1].y, lane));
    float z128 = 0.f;
    if (q == 0) { const c32 z = fn[posN(128)]; z128 = z.x * z.x + z.y * z.y; }
    wave_sync();                                         // every lane holds its Z values: the scratch may be overwritten
    float* pw = reinterpret_cast<float*>(sc) + f * kPowStride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = q + 16 * i;
        const c32 ptop = i == 0 ? (q == 0 ? k01[0].xy : prev0) : (q == 0 ? prev0 : prev1);
        const c32 zk[4] = {k01[i].xy, k01[i].zw, k23[i].xy, k23[i].zw};
        const c32 zp[4] = {ptop, p23[i].zw, p23[i].xy, p01[i].zw};
        const c32 ww[4] = {w01[i].xy, w01[i].zw, w23[i].xy, w23[i].zw};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const c32 P = add_conj(zk[e], zp[e]), Q = sub_conj(zk[e], zp[e]);
            const c32 wQ = cmul(Q, ww[e]);
            const c32 m2 = mag2(xy_re(P, wQ), xy_im(P, wQ));         // (|2 X[k]|^2, |2 X[256-k]|^2)
            pw[4 * b + e] = 0.25f * m2.x;
            pw[256 - 4 * b - e] = 0.25f * m2.y;                      // (k = 0: X[256], written by lane 0 only)
        }
    }
    if (q == 0) pw[128] = z128;                            // X[128] = conj(Z[128]); after the loop: b = 32 is nobody's
    for (int k = 257 + q; k < kPowStride; k += 16) pw[k] = 0.f;
    wave_sync();
}

__global__ __launch_bounds__(512) void k_logmel(MelParams p) {
    alignas(16) __shared__ c32 sc[8 * kWaveScratch];
    __shared__ float res[kMelMaxBands * 32];                // [n_mels][16 frames][2 ears]
    __shared__ float s_win[kNfft];
    __shared__ c32 s_tw512[kTw512Lds];
    alignas(16) __shared__ float s_w[kMelTableFloats];
    __shared__ int s_start[kMelMaxBands];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int groups = (p.n_frames + kSegFrames - 1) / kSegFrames, chunks = (groups + p.gpw - 1) / p.gpw;
    const int unit = blockIdx.x / chunks, g0 = (blockIdx.x % chunks) * p.gpw, g1 = min(groups, g0 + p.gpw);
    const int ch = wv >> 2, tbl = wv & 3;
    const float* row0 = p.x + (size_t)unit * 2 * p.len;
    f32x4* seg4 = reinterpret_cast<f32x4*>(sc);
    const float* seg = reinterpret_cast<const float*>(sc);
    SpecParams sp;                                          // the staging helper only needs these fields
    sp.x = p.x; sp.len = p.len; sp.pad_mode = p.pad_mode;
    f32x4 r[3];
    spec_seg_load(sp, row0, g0, t, r);
    s_win[t] = p.tb.win[t];
    if (t < 256) s_tw512[posN(t)] = p.tb.tw512[t];
    for (int e = t; e < p.n_mels * p.max_len; e += 512) s_w[e] = p.w[e];
    if (t < p.n_mels) s_start[t] = p.start[t];
    c32 wq = p.tb.twM[64 * (lane & 15)];
    // waited for HERE: if this load were still "pending" for the compiler inside the loop, its first use there would
    // get an s_waitcnt vmcnt(0), which (one in-order counter) also waits for the next segment's prefetch
    SSK_OPAQUE2(wq);
    for (int g = g0; g < g1; ++g) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (t + 512 * k < 2 * kSegQuads) seg4[t + 512 * k] = r[k];
        lds_barrier();
        const int fl = 4 * tbl + (lane >> 4);               // frame of this lane within the group
        const bool live = kSegFrames * g + fl < p.n_frames;
        c32 x[16];
        {
            const c32* y2 = reinterpret_cast<const c32*>(seg + ch * kSegLen + kHop * fl) + (lane & 15);
            const c32* w2 = reinterpret_cast<const c32*>(s_win) + (lane & 15);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const c32 s = lds_ld(y2 + 16 * j), w = lds_ld(w2 + 16 * j);
                x[j] = live ? mk2(w.x * s.x, w.y * s.y) : mk2(0.f, 0.f);
            }
        }
        lds_barrier();
        if (g + 1 < g1) spec_seg_load(sp, row0, g + 1, t, r);
        if (kSegFrames * g + 4 * tbl < p.n_frames) {        // wave-uniform: at least one live frame in this block
            c32* wsc = sc + wv * kWaveScratch;
            stft_power(wsc, lane, wq, s_tw512, x);
            const float* pw = reinterpret_cast<const float*>(wsc) + (lane >> 4) * kPowStride;
            for (int j = lane & 15; j < p.n_mels; j += 16) {
                // band starts and max_len are multiples of 4 (ABI contract): two aligned 16-byte LDS reads per 4 bins
                const f32x4* pj = reinterpret_cast<const f32x4*>(pw + s_start[j]);
                const f32x4* wj = reinterpret_cast<const f32x4*>(s_w + j * p.max_len);
                float acc = 0.f;
                for (int i = 0; i < (p.max_len >> 2); ++i) {
                    const f32x4 a = wj[i], b = pj[i];
                    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
                    acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                }
#if defined(__HIP_DEVICE_COMPILE__)
                const float v = __builtin_amdgcn_logf(acc + p.eps) * 0.69314718055994531f;
#else
                const float v = logf(acc + p.eps);
#endif
                res[(j * kSegFrames + fl) * 2 + ch] = v;
            }
        }
        lds_barrier();
        // rows of 16 frames x 2 ears = 128 contiguous bytes of out[unit][j][16 g ..][.]
        const int nf = min(kSegFrames, p.n_frames - kSegFrames * g);
        float* o = p.out + ((size_t)unit * p.n_mels * p.n_frames + kSegFrames * g) * 2;
        for (int e = t; e < p.n_mels * 32; e += 512) {
            const int j = e >> 5, c = e & 31;
            if (c < 2 * nf) o[(size_t)j * p.n_frames * 2 + c] = res[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_gccphat (EXTENSION, not in the reference: BASELINE configs[4] "GCC-PHAT"): per STFT frame
//   G[k] = X_l[k] conj(X_r[k]),  g = irfft(G / (|G| + eps), 512),  out[n][i][tf] = g[(i - max_lag) mod 512].
// 256 threads = 4 waves; a workgroup owns 16 frames of one unit and BOTH ears (the cross-spectrum needs them in one
// place), a wave its 4 frames: two forward 256-point FFTs (the packed real FFTs of the two ears), Hermitian split of
// both, normalised cross-spectrum, Hermitian merge, one inverse 256-point FFT = the 512 lags as packed pairs.  All
// stages reuse the STFT building blocks (radix-16 passes, padded natural-order layout, DPP for the mirror top bin);
// everything between the staged input segment and the result rows stays in registers / the wave's LDS scratch.
struct GccParams {
    const float* x;        // [N][2][len]
    float* out;            // [N][2*max_lag+1][n_frames]
    Tables tb;
    int len, n_frames, pad_mode, gpw, max_lag;
    float eps;
};
constexpr int kGccMaxLag = 32;
constexpr int kGccWaveScratch = 4 * kNatStride + 4 * kFrameStride;   // region A (natural order) + region B (tiles)

// 256-point forward FFT of 4 packed frames held as x[j] = frame[q + 16 j]; natural-order result to fn[] (padded layout);
// tile = transpose scratch
__device__ __forceinline__ void frame_fft_fwd(c32* tile, c32* nat, int lane, c32 wq, c32 (&x)[16]) {
    const int f = lane >> 4, q = lane & 15;
    c32* fr = tile + f * kFrameStride;
    c32* fn = nat + f * kNatStride;
    fft16<false>(x);
    SSK_OPAQUE2(wq);
    twiddle16<false>(x, wq);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_st(fr + r * 17 + q, x[r]);
    wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = lds_ld(fr + q * 17 + r);
    fft16<false>(x);
#pragma unroll
    for (int s = 0; s < 16; ++s) fn[q + posN(16 * s)] = x[s];
    wave_sync();
}

// this lane's 8 Hermitian bin pairs (k = 4b+e, 256-k), b = q + 16 i, of one ear: X2[k] and conj(X2[256-k])
struct EarQuads { f32x4 k01[2], k23[2], p01[2], p23[2]; };
__device__ __forceinline__ void ear_load(const c32* nat, int lane, EarQuads& z) {
    const int f = lane >> 4, q = lane & 15;
    const c32* fn = nat + f * kNatStride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = q + 16 * i;
        const f32x4* zk4 = reinterpret_cast<const f32x4*>(fn + posN(4 * b));
        const f32x4* zp4 = reinterpret_cast<const f32x4*>(fn + posN(252 - 4 * b));
        z.k01[i] = zk4[0]; z.k23[i] = zk4[1]; z.p01[i] = zp4[0]; z.p23[i] = zp4[1];
    }
}
__device__ __forceinline__ void ear_pairs(const EarQuads& z, int i, int lane, c32 (&zk)[4], c32 (&zp)[4]) {
    const int q = lane & 15;
    const c32 prev0 = mk2(row_ror1(z.p01[0].x, lane), row_ror1(z.p01[0].y, lane));
    const c32 prev1 = mk2(row_ror1(z.p01[1].x, lane), row_ror1(z.p01[1].y, lane));
    const c32 ptop = i == 0 ? (q == 0 ? z.k01[0].xy : prev0) : (q == 0 ? prev0 : prev1);
    zk[0] = z.k01[i].xy; zk[1] = z.k01[i].zw; zk[2] = z.k23[i].xy; zk[3] = z.k23[i].zw;
    zp[0] = ptop; zp[1] = z.p23[i].zw; zp[2] = z.p23[i].xy; zp[3] = z.p01[i].zw;
}

// a / (|a| + eps): the PHAT weighting
__device__ __forceinline__ c32 phat(c32 a, float eps) {
    const float m = fast_sqrt(a.x * a.x + a.y * a.y) + eps;
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(m);
#else
    const float r = 1.0f / m;
#endif
    return mk2(a.x * r, a.y * r);
}

__global__ __launch_bounds__(256) void k_gccphat(GccParams p) {
    alignas(16) __shared__ c32 sc[4 * kGccWaveScratch];     // staging (2 x 2912 floats), then 4 wave scratches
    __shared__ float res[(2 * kGccMaxLag + 1) * kSegFrames];
    __shared__ float s_win[kNfft];
    __shared__ c32 s_tw512[kTw512Lds];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, q = lane & 15;
    const int groups = (p.n_frames + kSegFrames - 1) / kSegFrames, chunks = (groups + p.gpw - 1) / p.gpw;
    const int unit = blockIdx.x / chunks, g0 = (blockIdx.x % chunks) * p.gpw, g1 = min(groups, g0 + p.gpw);
    const int n_lags = 2 * p.max_lag + 1;
    const float* row0 = p.x + (size_t)unit * 2 * p.len;
    f32x4* seg4 = reinterpret_cast<f32x4*>(sc);
    const float* seg = reinterpret_cast<const float*>(sc);
    SpecParams sp;
    sp.x = p.x; sp.len = p.len; sp.pad_mode = p.pad_mode;
    for (int e = t; e < kNfft; e += 256) s_win[e] = p.tb.win[e];
    s_tw512[posN(t)] = p.tb.tw512[t];
    c32 wq = p.tb.twM[64 * q];
    SSK_OPAQUE2(wq);                                        // see k_spectrogram
    const float eps4 = 4.f * p.eps;                         // the split yields 2X, so the products carry a factor 4
    for (int g = g0; g < g1; ++g) {
        // stage the two padded segments (same helper as k_spectrogram, written for 512 threads: two half rounds)
        f32x4 r[3];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            spec_seg_load(sp, row0, g, 2 * t + half, r);    // quads 2t+half + 512 k  (a permutation of 0..1535)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (2 * t + half + 512 * k < 2 * kSegQuads) seg4[2 * t + half + 512 * k] = r[k];
        }
        lds_barrier();
        const int fl = 4 * wv + (lane >> 4);
        const bool live = kSegFrames * g + fl < p.n_frames;
        c32 xl[16], xr[16];
        {
            const c32* w2 = reinterpret_cast<const c32*>(s_win) + q;
            const c32* yl = reinterpret_cast<const c32*>(seg + kHop * fl) + q;
            const c32* yr = reinterpret_cast<const c32*>(seg + kSegLen + kHop * fl) + q;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const c32 w = lds_ld(w2 + 16 * j), a = lds_ld(yl + 16 * j), b = lds_ld(yr + 16 * j);
                xl[j] = live ? mk2(w.x * a.x, w.y * a.y) : mk2(0.f, 0.f);
                xr[j] = live ? mk2(w.x * b.x, w.y * b.y) : mk2(0.f, 0.f);
            }
        }
        lds_barrier();                                      // staging area dead
        if (kSegFrames * g + 4 * wv < p.n_frames) {         // wave-uniform
            c32* regA = sc + wv * kGccWaveScratch;          // natural-order spectra / V
            c32* regB = regA + 4 * kNatStride;              // transpose tiles
            EarQuads zl, zr;
            frame_fft_fwd(regB, regA, lane, wq, xl);
            ear_load(regA, lane, zl);
            c32 z128l = regA[(lane >> 4) * kNatStride + posN(128)];
            wave_sync();                                    // left spectrum is in registers: region A is free again
            frame_fft_fwd(regB, regA, lane, wq, xr);
            ear_load(regA, lane, zr);
            const c32 z128r = regA[(lane >> 4) * kNatStride + posN(128)];
            f32x4 w01[2], w23[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4* w4 = reinterpret_cast<const f32x4*>(s_tw512 + posN(4 * (q + 16 * i)));
                w01[i] = w4[0]; w23[i] = w4[1];
            }
            wave_sync();                                    // every lane holds its bins: V may overwrite region A
            c32* vn = regA + (lane >> 4) * kNatStride;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int b = q + 16 * i;
                c32 lk[4], lp[4], rk[4], rp[4];
                ear_pairs(zl, i, lane, lk, lp);
                ear_pairs(zr, i, lane, rk, rp);
                const c32 ww[4] = {w01[i].xy, w01[i].zw, w23[i].xy, w23[i].zw};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // Hermitian split of both ears: X = 2 X[k], Y = conj(2 X[256-k])
                    const c32 Pl = add_conj(lk[e], lp[e]), Ql = sub_conj(lk[e], lp[e]), wl = cmul(Ql, ww[e]);
                    const c32 Pr = add_conj(rk[e], rp[e]), Qr = sub_conj(rk[e], rp[e]), wr = cmul(Qr, ww[e]);
                    const c32 Xl = add_mi(Pl, wl), Yl = add_pi(Pl, wl), Xr = add_mi(Pr, wr), Yr = add_pi(Pr, wr);
                    c32 gk = phat(cmulc(Xl, Xr), eps4);                       // G[k]
                    const c32 gpc = phat(cmulc(Yl, Yr), eps4);                // conj(G[256-k])
                    c32 gp = mk2(gpc.x, -gpc.y);
                    herm_inv(gk, gp, ww[e]);                                  // -> 2 V[k], 2 V[256-k]
                    vn[posN(4 * b + e)] = gk;
                    if (4 * b + e != 0) vn[posN(256 - 4 * b - e)] = gp;       // V[256] does not exist
                }
            }
            if (q == 0) {                                   // k = 128 pairs with itself: X[128] = conj(Z[128])
                const c32 g128 = phat(cmulc(mk2(z128l.x, -z128l.y), mk2(z128r.x, -z128r.y)), p.eps);
                vn[posN(128)] = mk2(2.f * g128.x, -2.f * g128.y);             // 2 V[128] = 2 conj(G[128])
            }
            wave_sync();
            // inverse 256-point FFT of V: lane q holds V[q + 16 j]; result x[s] = 512 * (g[2n], g[2n+1]), n = q + 16 s
            c32 x[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = lds_ld(vn + q + posN(16 * j));
            fft16<true>(x);
            c32 wqi = wq;
            SSK_OPAQUE2(wqi);
            twiddle16<true>(x, wqi);
            c32* fr = regB + (lane >> 4) * kFrameStride;
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) fr[r2 * 17 + q] = x[r2];
            wave_sync();
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) x[r2] = lds_ld(fr + q * 17 + r2);
            fft16<true>(x);
            constexpr float inv = 1.0f / 512.0f;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tau = 2 * (q + 16 * s2) + u;                    // lag 0..511; >= 256 means tau - 512
                    const int i = tau <= p.max_lag ? tau + p.max_lag : tau >= kNfft - p.max_lag ? tau - kNfft + p.max_lag : -1;
                    if (i >= 0) res[i * kSegFrames + fl] = inv * (u ? x[s2].y : x[s2].x);
                }
            }
        }
        lds_barrier();
        const int nf = min(kSegFrames, p.n_frames - kSegFrames * g);
        float* o = p.out + (size_t)unit * n_lags * p.n_frames + kSegFrames * g;
        for (int e = t; e < n_lags * kSegFrames; e += 256) {
            const int i = e >> 4, c = e & 15;
            if (c < nf) o[(size_t)i * p.n_frames + c] = res[e];
        }
        lds_barrier();                                      // res and the scratch are reused by the next group
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv: one workgroup per (unit, ear, output block j = blockIdx.y).
// unit descriptor desc[n][8], two terms (source, distractor):
//   desc[4k+0] = RIR bank index (-1: term absent)      desc[4k+1] = spectrum slot of window m = m_min
//   desc[4k+2] = m_min                                  desc[4k+3] = number of stored windows
// Silent units (simulator.py:610-612) have both terms absent and produce exact zeros, as do empty RIRs.
// Y_j = sum over (term, RIR block i) of H_i * S_{j-i}; one accumulator per workgroup, so multi-block rows
// (44.1 kHz: 3 output blocks) re-run the forward FFTs per output block instead of holding 3 accumulators
// (96 VGPRs) that a 1024-thread workgroup does not have.
constexpr int kMaxBuckets = 4;
constexpr int kTabUnits = 256;
// words per unit of the kernel-argument table: {bank index | -1, window-spectrum slot, OUTPUT unit}.  The host deals the
// units out sorted by window spectrum (one group of sounds per XCD: row_slot), so the rows that read one 128-KiB spectrum sit
// on one XCD and find it in that L2; word 2 says where the row's results go (the caller's unit order is untouched)
constexpr int kTabWords = 3;
struct BankBucket {
    const float* rir;            // planar [n, 2, cap] rows, zero beyond each entry's length
    const f32x4* hspec;          // spectral form [n][2][h_blocks][8192] f32x4, or nullptr
    int first;                   // global bank index of the bucket's entry 0
    int cap;                     // samples per (entry, ear) row (even)
    int h_blocks;                // ceil(cap / kB)
    int pad_;
};
struct ConvParams {
    const f32x4* spec;          // [slots][8192] f32x4 (kernel order, see k_source_windows)
    const float* rir;            // RIR bank
    const int* rir_len;          // [R]
    const int* desc;             // [N][8]
    float* out;                  // audiogoal [N][2][out_len] or nullptr
    float* sgram;                // spectrogram [N][65][T4][2] or nullptr (FUSE only)
    Tables tb;
    long long rir_unit_stride;   // floats between bank entries
    int rir_chan_stride;         // floats between ears      (planar [R,2,L]: L ; wav [R,L,2]: 1)
    int rir_elem_stride;         // floats between samples   (planar: 1 ; wav: 2)
    int rir_cap;                 // samples readable per (entry, ear); rows are ZERO beyond rir_len[r] up to rir_cap
    int n_valid;                 // samples computed per row (<= nb_y*kB); [n_valid, out_len) is zero-filled
    int out_len;                 // row length
    int n_frames, t4, pad_mode;  // spectrogram geometry for the fused path
    int fade_len;                // XFADE kernels: cross-fade ramp covers samples 0..fade_len (continuous_simulator.py:47-53)
    // spectral RIR bank (k_conv_spec): block spectra H' = 2*rFFT_{2kB}(rir[i*kB:(i+1)*kB]) in kernel order,
    // [entry][ear][h_blocks][8192] f32x4; rir / rir_*_stride / rir_cap are unused by those kernels
    const f32x4* hspec;
    int h_blocks;
    int xcd_map;                 // != 0: launch slots are dealt to the XCDs in contiguous ranges (row_slot)
    int nb_y;                    // output blocks per row; k_conv_spec: grid = 2N * nb_y, slot = (row, j), j fastest
    // Small steps (fewer rows than CUs; the reference steps 5-10 envs per GPU: ss_baselines/av_nav/config/audionav/*/
    // train_telephone/audiogoal_depth_ddppo.yaml:3): a (unit, ear) row is rendered by 2^parts_log2 workgroups on as many CUs.
    // Each of them runs the whole convolution of the row (redundant, on CUs that would idle) and then ONLY ITS SHARE of the
    // pooled STFT blocks, which are independent of each other: the STFT phase - 26 blocks on 16 waves = two rounds at four
    // waves per SIMD, VALU-bound - becomes one round of <= 13 / 7 / 4 blocks.  Fused one-block kernels only; 0 = one
    // workgroup per row.
    int parts_log2;
    // k_obs_rows (rows longer than one block, fused): per-workgroup scratch for the block spectra H' of the row being
    // rendered, [gridDim.x][stash_terms][stash_nbh][8192] f32x4 (time-domain bank only); stash_terms = 1 when the
    // launch has no distractor terms
    f32x4* stash;
    int stash_nbh, stash_terms;
    int n_terms;                 // k_obs_rows: descriptor terms read per unit (1 under SS_FLAG_NO_DISTRACTOR)
    // Length-bucketed bank: bank entries [bk[b].first, next bucket's first) live in bucket b + 1, an
    // allocation of its own with its own row capacity; entries below bk[0].first are bucket 0 = the fields above (rir /
    // rir_*_stride / rir_cap, hspec / h_blocks).  n_buckets = 1: one bank, as before.  The loop-free kernels (SIMPLE,
    // k_conv_rows, k_conv_spec_rows) only ever see bucket-0 entries (the launcher checks); the loop kernels resolve the
    // bucket per term from the wave-uniform bank index (scalar compares, no memory access).
    int n_buckets;
    BankBucket bk[kMaxBuckets - 1];
#if defined(SS_LADDER)
    int dbg;                     // timing experiments only (scripts/gpu_ladder.sh, -DSS_LADDER builds): early exit point
#endif
};

// Loop-free kernels, launches of <= kTabUnits units whose descriptors the HOST knows (the context API): the two words a
// SIMPLE row needs - {bank index | -1, window-spectrum slot} - ride in the kernel-argument block itself, so the chain in
// front of the row's first loads is ONE scalar fetch (arguments) instead of arguments -> descriptor (which for the
// context's in-place descriptors is a round trip over the host link) -> length word.  A kernel argument of its OWN, passed
// to the TAB instantiations only (k_conv<.., SIMPLE, .., TAB>, k_conv_spec<.., SIMPLE, TAB>, k_conv32<.., TAB>): every
// other launch keeps a kernarg block of a few hundred bytes (inside ConvParams the 2 KiB went up with every
// launch of every kernel and sat next to the 4 KiB kernarg limit).
template <bool TAB>
struct UnitTab { };
template <>
struct UnitTab<true> { int tab[kTabWords * kTabUnits]; };
static_assert(sizeof(ConvParams) <= 512, "ConvParams is passed by value to every conv kernel: keep it small");
static_assert(sizeof(ConvParams) + sizeof(UnitTab<true>) + 64 <= 4096, "kernel arguments of the TAB instantiations");

// Workgroup b of a 1-D launch runs on XCD b % 8 (MI355X_MICROARCH: observed dispatch order, for speed only).  Dealing
// the (unit, ear) rows out in that order puts the two ears of a unit - which read the SAME 128 KiB window spectrum -
// on two different XCDs, i.e. in two different L2s, and scatters the units of one sound over all eight.  row_slot()
// gives XCD x the contiguous slot range [x*G/8, (x+1)*G/8): both ears of a unit, and (with units sorted by sound by the
// planner) all rows of a group of sounds, share one L2.
__device__ __forceinline__ int row_slot(int b, int G, int on) {
    if (!on) return b;
    const int x = b & 7, q = G >> 3, r = G & 7;
    return x * q + min(x, r) + (b >> 3);
}

// Time-domain row of bank entry ridx, ear ch: address, row capacity and element stride (wave-uniform).
struct BankRow { const float* h; int cap, es; };
// BUCKETS = false: the launch's bank is ONE allocation (n_buckets == 1: the launcher checks) - the three bucket descriptors
// (24 scalar registers) and the compare chains never enter the kernel (k_obs_rows: 81-118 -> SGPR spills)
template <bool BUCKETS = true>
__device__ __forceinline__ BankRow bank_row(const ConvParams& p, int ridx, int ch) {
    BankRow r{p.rir + (size_t)ridx * p.rir_unit_stride + (size_t)ch * p.rir_chan_stride, p.rir_cap, p.rir_elem_stride};
    if (!BUCKETS) return r;
#pragma unroll
    for (int b = 0; b < kMaxBuckets - 1; ++b)
        if (b + 1 < p.n_buckets && ridx >= p.bk[b].first) {
            r.h = p.bk[b].rir + ((size_t)(ridx - p.bk[b].first) * 2 + ch) * p.bk[b].cap;
            r.cap = p.bk[b].cap;
            r.es = 1;
        }
    return r;
}
// Spectral row of bank entry ridx, ear ch: first block spectrum and the number of blocks stored per row.
struct BankSpec { const f32x4* hp; int h_blocks; };
template <bool BUCKETS = true>
__device__ __forceinline__ BankSpec bank_spec(const ConvParams& p, int ridx, int ch) {
    BankSpec r{p.hspec + ((size_t)ridx * 2 + ch) * (size_t)p.h_blocks * (kSpecComplex / 2), p.h_blocks};
    if (!BUCKETS) return r;
#pragma unroll
    for (int b = 0; b < kMaxBuckets - 1; ++b)
        if (b + 1 < p.n_buckets && ridx >= p.bk[b].first) {
            r.hp = p.bk[b].hspec + ((size_t)(ridx - p.bk[b].first) * 2 + ch) * (size_t)p.bk[b].h_blocks * (kSpecComplex / 2);
            r.h_blocks = p.bk[b].h_blocks;
        }
    return r;
}

// forward FFT of RIR block i of one ear + multiply by the window spectrum `slot` -> acc (= or +=)
template <bool ACCUMULATE, bool PREFETCH>
__device__ __forceinline__ void conv_block(c32* lds, const ConvParams& p, const ThreadTw& tw, int t, const BankRow& br,
                                           int L, int i, int slot, c32 (&acc)[2][8]) {
    const float* h = br.h;
    // PREFETCH = issue the 8 window-spectrum loads (L2/MALL hits) at the start of pass 3 instead of at the item
    // stage, so their latency hides under pass 3.  (Issuing them before pass 1 was measured SLOWER, +1.2 us:
    // 128 KB of L2 reads queue ahead of the RIR's HBM loads on the in-order vmcnt path.)  Only the loop-free
    // kernel can afford the 32 VGPRs; the looped kernel also carries an accumulator across the passes.
    const f32x4* sp = p.spec + (size_t)slot * (kSpecComplex / 2) + t;
    f32x4 sv[2][4];
    // bank rows are zero-padded to rir_cap, so the only bound is the row capacity (L is used for block counts)
    const int lo = i * kB, es = br.es, cap = br.cap;
    (void)L;
    if (es == 1 && !(cap & 1) && !(reinterpret_cast<size_t>(h) & 7)) {       // planar, 8-byte aligned rows
        const c32* h2 = reinterpret_cast<const c32*>(h + lo);
        const int m_end = (cap - lo) >> 1;
        pass1_fwd<true>(lds, tw.p1, t, [&](int m) { return m < m_end ? ld_stream(h2 + m) : mk2(0.f, 0.f); });
    } else {
        pass1_fwd<true>(lds, tw.p1, t, [&](int m) {
            const int n = lo + 2 * m;
            return mk2(n < cap ? h[(size_t)n * es] : 0.f, n + 1 < cap ? h[(size_t)(n + 1) * es] : 0.f);
        });
    }
    lds_barrier();
    pass2<false>(lds, tw.p2, t);
    lds_barrier();
    if (PREFETCH) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) sv[s][hh] = sp[(s * 4 + hh) * 1024];
    }
    pass3_fwd(lds, t);
    lds_barrier();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (!PREFETCH) {
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) sv[s][hh] = sp[(s * 4 + hh) * 1024];
        }
        c32 v[8];
        item_load_fwd(lds, s ? tw.i1 : tw.i0, t + 1024 * s, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const c32 w = (e & 1) ? sv[s][e >> 1].zw : sv[s][e >> 1].xy;
            c32 pr = cmul(v[e], w);
            if (s == 0 && e == 0 && t == 0) pr = mk2(v[0].x * w.x, v[0].y * w.y);   // (X[0], X[16384]) are real
            if (ACCUMULATE) acc[s][e] += pr;
            else acc[s][e] = pr;
        }
    }
}

// One (unit, ear) row of the SIMPLE case after pass 1 has filled the LDS buffer: passes 2-3, the item stage, passes
// 3'-2'-1'.  The item stage is IN PLACE per thread (a thread writes back exactly the layout-B slots it read), so each
// item is carried through load -> Hermitian split -> multiply by the window spectrum -> merge -> inverse radix-4 ->
// store before the next one starts: no accumulator array, no barrier between the two halves, ~35 fewer live VGPRs.
// Item 0's spectrum values are fetched before pass 3 and item 1's while item 0 is being processed, so neither L2
// round trip is exposed.
__device__ __forceinline__ void multiply_item(c32* lds, c32 wg, int t, int s, const f32x4 (&sv)[4]) {
    c32 v[8];
    item_load_fwd(lds, wg, t + 1024 * s, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const c32 w = (e & 1) ? sv[e >> 1].zw : sv[e >> 1].xy;
        c32 pr = cmul(v[e], w);
        if (s == 0 && e == 0 && t == 0) pr = mk2(v[0].x * w.x, v[0].y * w.y);   // (X[0], X[16384]) are real
        v[e] = pr;
    }
    item_store_inv(lds, wg, t + 1024 * s, v);
}

// forward half: passes 2-3 and the in-place item stage (LDS ends up holding the product spectrum in layout B)
__device__ __forceinline__ void simple_row_fwd(c32* lds, const ConvParams& p, const ThreadTw& tw, int t, int slot) {
    const f32x4* sp = p.spec + (size_t)slot * (kSpecComplex / 2) + t;
    lds_barrier();
    pass2<false>(lds, tw.p2, t);
    lds_barrier();
    f32x4 sv0[4], sv1[4];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) sv0[hh] = sp[hh * 1024];
    pass3_fwd(lds, t);
    lds_barrier();
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) sv1[hh] = sp[(4 + hh) * 1024];
    multiply_item(lds, tw.i0, t, 0, sv0);
    multiply_item(lds, tw.i1, t, 1, sv1);
}

// inverse half: passes 3'-2'-1' -> the row's last kB samples in registers
__device__ __forceinline__ void simple_row_inv(c32* lds, const ThreadTw& tw, int t, c32 (&y)[8]) {
    lds_barrier();
    pass3_inv(lds, t);
    lds_barrier();
    pass2<true>(lds, tw.p2, t);
    lds_barrier();
    pass1_inv(lds, tw.p1, t, y);
}

__device__ __forceinline__ void simple_row_after_pass1(c32* lds, const ConvParams& p, const ThreadTw& tw, int t, int slot,
                                                       c32 (&y)[8]) {
    simple_row_fwd(lds, p, tw, t, slot);
    simple_row_inv(lds, tw, t, y);
}

// write one output block (kB samples starting at j*kB) of row `row`; block 0 also zero-fills [n_valid, out_len)
__device__ __forceinline__ void store_row_block(const ConvParams& p, int t, size_t row, int j, const c32 (&y)[8]) {
    if (!p.out) return;
    float* orow = p.out + row * p.out_len + j * kB;
    const int nv = p.n_valid - j * kB;                     // valid samples of this block
    if (!(nv & 1) && !(reinterpret_cast<size_t>(orow) & 7)) {
        c32* o2 = reinterpret_cast<c32*>(orow) + t;
        const int m_end = nv >> 1;
#pragma unroll
        for (int a = 0; a < 8; ++a) if (t + 1024 * a < m_end) st_stream(o2 + 1024 * a, y[a]);
    } else {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int n = 2 * (t + 1024 * a);
            if (n < nv) orow[n] = y[a].x;
            if (n + 1 < nv) orow[n + 1] = y[a].y;
        }
    }
    if (j == 0) for (int n = p.n_valid + t; n < p.out_len; n += kT) p.out[row * p.out_len + n] = 0.f;
}

// Fused STFT phase (out_len <= kB, t4 <= 26): the 1-s row goes from registers into LDS and feeds the STFT directly.
// WIDE (round 4): rows LONGER than one block of which only block 0 is rendered (n_valid <= kB, out_len > kB: every SoundSpaces
// 2.0 step at 44.1 kHz, continuous_simulator.py:413-456 - 0.25 s of a 1-s row).  Everything behind n_valid is zero, so
// at most 26 pooled blocks are live (live_blocks) and they only need samples of block 0; the other t4 - live columns are
// written as zeros.  One launch, no waveform buffer, block spectra accumulated in registers (each is used once).
constexpr int kResFloats = kBins4 * 26;     // pooled spectrogram of one ear (<= 26 live blocks on the fused path)
// pooled blocks [part * per, (part + 1) * per) of a row belong to workgroup `part` of its 2^parts_log2 (ConvParams::parts_log2)
__host__ __device__ constexpr int part_blocks(int t4, int parts_log2) { return (t4 + (1 << parts_log2) - 1) >> parts_log2; }

template <bool WIDE = false>
__device__ __forceinline__ void fused_stft_phase(c32* lds, const ConvParams& p, int t, int unit, int ch, const c32 (&y)[8],
                                                 const float* s_win, const c32* s_tw512, c32 wq, float* s_res, int part = 0) {
    // The row goes to LDS with librosa's centre padding materialised around it (256 samples on each side), so that
    // every frame is an aligned, branch-free read.  (With the padding resolved per sample at load time, the three
    // waves that own the first / last frames ran a ~300-instruction edge path on top of their two blocks; two of them
    // share SIMD 0, which made the whole phase ~20 % longer than its arithmetic.)
    float* yl = reinterpret_cast<float*>(lds) + kNfft / 2;          // sample 0 of the row
    const int len = p.out_len;
    lds_barrier();   // all pass-1' reads of layout A are done
    c32* yl2 = reinterpret_cast<c32*>(yl) + t;            // the row as packed pairs: one ds_write_b64 per pair
    // split rows (parts_log2 > 0): only the 2048-sample chunks this workgroup's pooled blocks read - block b reads samples
    // [640 b - 256, 640 b + 736) - plus the chunk the centre padding mirrors (first / last part); wave-uniform tests
    const int per_w = WIDE ? 26 : part_blocks(p.t4, p.parts_log2);
    const int s_lo = WIDE ? 0 : kHop * kPool * (part * per_w) - kNfft / 2;
    const int s_hi = WIDE || p.parts_log2 == 0 ? 2 * kB : kHop * kPool * (part * per_w + per_w - 1) + kHop * (kPool - 1) + kNfft / 2;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        if (2048 * (a + 1) <= s_lo || 2048 * a >= s_hi) continue;
        const int n = 2 * (t + 1024 * a);
        yl2[1024 * a] = mk2(n < p.n_valid ? y[a].x : 0.f, n + 1 < p.n_valid ? y[a].y : 0.f);   // zeros beyond n_valid
    }
    lds_barrier();
    if (t < kNfft / 2) {                                  // the two pads: reflect (excluding the edge sample) or zeros
        const bool refl = p.pad_mode == 0;
        const float l = refl ? yl[1 + t] : 0.f;
        yl[-1 - t] = l;
        if (WIDE) {                                         // the row goes on (as zeros) behind the block: no right edge here;
            yl[kB + t] = 0.f;                               // the last frame of live block 25 (n_valid > 15744) reads up to
            yl[kB + kNfft / 2 + t] = 0.f;                   // sample kB + 352
        }
        else yl[len + t] = refl ? yl[len - 2 - t] : 0.f;
    }
    lds_barrier();
    const int lane = t & 63, wv = t >> 6;
    // The padded row (<= 16896 floats) sits at the start of the LDS buffer and the 16 waves' STFT scratch overlays the
    // whole buffer, so every wave first pulls the frames of BOTH its time blocks (wv and wv+16; t4 <= 26) into
    // registers; after one barrier the row is dead and each wave runs its two blocks back to back in its private
    // scratch with wave-scope synchronisation only (no workgroup barrier, waves free-running).
    c32 x0[16], x1[16];
    // Short steps (SS2.0: 0.25 s of a 1-s row): pooled blocks that are exactly zero (live_blocks) are written, not computed
    // (16 kHz, 0.25 s: 7 live blocks of 26).  live == t4 otherwise.
    const int live = live_blocks(p.n_valid, len, p.t4);
    // this workgroup's share of the pooled blocks: [b_lo, b_lo + per) (one workgroup per row: all of them)
    const int per = WIDE ? 26 : part_blocks(p.t4, p.parts_log2), b_lo = WIDE ? 0 : part * per;
    const int b_end = min(live, b_lo + per);
    const int bw = b_lo + wv;                               // this wave's blocks: bw and bw + 16
    const bool one = bw < b_end, two = bw + 16 < b_end;
    const float* padded = reinterpret_cast<const float*>(lds);   // frame tf = floats [160 tf, 160 tf + 512)
#if defined(SS_NO_WINREG)                                          // (A/B arm: the window pairs read from LDS per block)
    stft_load_padded(padded, 4 * bw + (lane >> 4), one ? p.n_frames : 0, lane & 15, s_win, x0);
    stft_load_padded(padded, 4 * (bw + 16) + (lane >> 4), two ? p.n_frames : 0, lane & 15, s_win, x1);
#else
    {
        c32 w[16];
        stft_window_pairs(s_win, lane & 15, w);
        stft_load_padded_w(padded, 4 * bw + (lane >> 4), one ? p.n_frames : 0, lane & 15, w, x0);
        stft_load_padded_w(padded, 4 * (bw + 16) + (lane >> 4), two ? p.n_frames : 0, lane & 15, w, x1);
    }
#endif
    lds_barrier();
    // The pooled values of this ear are collected in LDS and leave together: written straight from the blocks, lane r
    // stores row r of out[unit][r][block][ear] - 64 lanes, 64 different cache lines, 4 bytes each, 26 times per workgroup.
    // From s_res, 32 threads per pooled row write its blocks: every other float of a contiguous range.
    const int rs = WIDE ? 26 : p.t4;                        // row stride of s_res
    if (one)
        stft_block(lds + wv * kWaveScratch, lane, wq, s_tw512, x0, [&](int b, float v) { s_res[b * rs + bw] = v; });
    if (two) {
        wave_sync();
        stft_block(lds + wv * kWaveScratch, lane, wq, s_tw512, x1, [&](int b, float v) { s_res[b * rs + bw + 16] = v; });
    }
    lds_barrier();
    float* o = p.sgram + (size_t)unit * kBins4 * p.t4 * 2 + ch;
    if (WIDE) {                                             // t4 columns per pooled row, the live ones from s_res
        for (int idx = t; idx < kBins4 * p.t4; idx += kT) {
            const int b = idx / p.t4, k = idx - b * p.t4;
            o[2 * idx] = k < live ? s_res[b * rs + k] : 0.f;
        }
        return;
    }
    const int k = b_lo + (t & 31);                          // (t4 <= 26 on this path)
    if ((t & 31) < per && k < p.t4)
        for (int b = t >> 5; b < kBins4; b += kT / 32) o[2 * (b * p.t4 + k)] = k < live ? s_res[b * p.t4 + k] : 0.f;
}

// SIMPLE: the caller guarantees one output block (nb_y == 1), RIR capacity <= kB and no distractor term,
// so a unit is at most ONE forward FFT: straight-line code, no accumulator carried across passes, no scratch.
// XFADE (SoundSpaces 2.0 CROSSFADE, continuous_simulator.py:47-53, 422-424): term 1 of the descriptor is not a
// distractor to be added but the PREVIOUS step's RIR: the row is convolved with it first, the first fade_len+1 samples
// of that result (at most two packed pairs per thread) are kept in registers, then the row is convolved with the
// current RIR (term 0) and the head of the row becomes prev*(fade-n)/fade + cur*n/fade.  One launch, and in the
// fused kernel the spectrogram is taken from the blended row without it ever leaving the CU.
template <bool FUSE, bool SIMPLE, bool XFADE = false, bool TAB = false, bool WIDE = false>
__global__ __launch_bounds__(1024) void k_conv(ConvParams p, UnitTab<TAB> ut = UnitTab<TAB>()) {
    static_assert(!(SIMPLE && XFADE), "the cross-fade needs the two-term loop kernel");
    static_assert(!WIDE || (FUSE && !SIMPLE), "WIDE: fused loop kernel for rows of which only block 0 is rendered");
    static_assert(!TAB || SIMPLE, "the unit table serves the loop-free kernel");
    __shared__ c32 lds[FUSE && 16 * kWaveScratch > kLdsComplex ? 16 * kWaveScratch : kLdsComplex];
    const int t = threadIdx.x;
    // grid (2N rows, nb_y output blocks).  (Putting the blocks of a row next to each other in slot order, as k_conv_spec
    // does, was measured 10 % SLOWER here at 44.1 kHz: 91 vs 82 us per 128 units.)
    const int slot = row_slot(blockIdx.x, gridDim.x, p.xcd_map);
    // fused one-block rows may be rendered by 2^parts_log2 workgroups each (ConvParams::parts_log2): slot = (row, part)
    constexpr bool PARTS = FUSE && !WIDE;
    const int part = PARTS ? slot & ((1 << p.parts_log2) - 1) : 0, row = PARTS ? slot >> p.parts_log2 : slot;
    if (PARTS && part && part * part_blocks(p.t4, p.parts_log2) >= p.t4) return;     // no pooled block left for this part
    const int unit = row >> 1, ch = row & 1, j = SIMPLE ? 0 : blockIdx.y;
    const int* d = p.desc + 8 * unit;
    const ThreadTw tw = load_thread_tw(p.tb.twM, p.tb.twItem, t);
    // fused path: the STFT's window and exp(-2 pi i k/512) tables are staged in the 21 KiB of LDS the FFT buffer
    // leaves free, and the lane's 256-point twiddle is fetched now, so the STFT phase starts no global loads
    __shared__ float s_win[FUSE ? kNfft : 1];
    __shared__ c32 s_tw512[FUSE ? kTw512Lds : 1];
    // The table values are only FETCHED here (three registers); they go to LDS after the convolution.  Staging them
    // right away put two load -> s_waitcnt vmcnt(0) -> ds_write round trips (~1.7 us) in front of the RIR loads.
    c32 wq = mk2(1.f, 0.f), tw512_v = mk2(0.f, 0.f);
    float win_v = 0.f;
    if (FUSE) {                                         // unconditional (clamped) loads: a predicated load is merged with
        win_v = p.tb.win[t & (kNfft - 1)];              // the default value by a register move, i.e. waited for right here;
        tw512_v = p.tb.tw512[t & 255];                  // only threads < 512 / < 256 store theirs to LDS
        wq = p.tb.twM[64 * (t & 15)];
    }

    c32 acc[2][8];
    c32 y[8];
    bool any = false;
    if (SIMPLE) {
        int ridx;
        if constexpr (TAB) ridx = ut.tab[kTabWords * unit]; else ridx = __builtin_amdgcn_readfirstlane(d[0]);
        bool active = false;
        if (ridx >= 0) {
            const float* h = p.rir + (size_t)ridx * p.rir_unit_stride + (size_t)ch * p.rir_chan_stride;
            const int es = p.rir_elem_stride, cap = p.rir_cap;
            const bool planar = es == 1 && !(cap & 1) && !(reinterpret_cast<size_t>(h) & 7);   // 8-byte aligned rows
            // the row's address needs only ridx: its loads go out BEFORE the length / window words are waited for
            c32 hraw[8];
            if (planar) {
                const c32* h2 = reinterpret_cast<const c32*>(h);
                const int m_end = cap >> 1;
#pragma unroll
                for (int a = 0; a < 8; ++a) hraw[a] = ld_stream(h2 + min(t + 1024 * a, m_end - 1));   // clamped, not
            }                                                   // predicated (see the table loads); masked where consumed
            // table form: the host has already folded "window 0 is stored" into the index, and an empty RIR needs no length
            // word: its row is zero up to the capacity, so its convolution comes out as exact zeros (as in k_conv_spec)
            int slot0 = 0;
            bool ok = true;
            if constexpr (TAB) {
                slot0 = ut.tab[kTabWords * unit + 1];
            } else {
                const int L = __builtin_amdgcn_readfirstlane(p.rir_len[ridx]);
                const int spec0 = __builtin_amdgcn_readfirstlane(d[1]);
                const int m_min = __builtin_amdgcn_readfirstlane(d[2]);
                const int m_cnt = __builtin_amdgcn_readfirstlane(d[3]);
                slot0 = spec0 - m_min;
                ok = L > 0 && m_min <= 0 && m_min + m_cnt > 0;
            }
            if (ok) {
                if (planar) {
                    const int m_end = cap >> 1;
                    pass1_fwd<true>(lds, tw.p1, t, [&](int m) { return m < m_end ? hraw[(m - t) >> 10] : mk2(0.f, 0.f); });
                } else {
                    pass1_fwd<true>(lds, tw.p1, t, [&](int m) {
                        const int n = 2 * m;
                        return mk2(n < cap ? h[(size_t)n * es] : 0.f, n + 1 < cap ? h[(size_t)(n + 1) * es] : 0.f);
                    });
                }
                simple_row_after_pass1(lds, p, tw, t, slot0, y);
                active = true;
            }
        }
        if (!active) {
#pragma unroll
            for (int a = 0; a < 8; ++a) y[a] = mk2(0.f, 0.f);
        }
    }
    // XFADE: the head (samples 0..fade_len <= 2 kPrevPairs - 2) of the row convolved with the previous RIR waits here, in the LDS
    // the FFT buffer leaves free, while the row is convolved with the current RIR: kept in registers it spilled
    // (the loop kernel already carries the accumulator across the passes at the 128-VGPR cap)
    __shared__ c32 s_prev[XFADE ? kPrevPairs : 1];
    // results of the fused STFT phase (see fused_stft_phase); the XFADE kernels are at the LDS limit and reuse s_prev,
    // which is dead once the row is blended
    __shared__ float s_res_own[FUSE && !XFADE ? kResFloats : 1];
    static_assert(2 * kPrevPairs >= kResFloats, "s_prev must be able to hold the pooled spectrogram");
    float* s_res = XFADE ? reinterpret_cast<float*>(s_prev) : s_res_own;
    bool have_prev = false;
    if (!SIMPLE) {
        // XFADE: round 0 = term 1 alone (previous RIR; only block 0 holds ramp samples), round 1 = term 0.
        // The two rounds are two inlined copies of the body; each works on its own opaque copy of the lane id so that
        // nothing lane-invariant (addresses, twiddle chains) is shared between the copies and kept alive across them.
#pragma unroll
        for (int round = XFADE ? 0 : 1; round < 2; ++round) {
            const int term_lo = XFADE ? 1 - round : 0, term_hi = XFADE ? term_lo + 1 : 2;
            if (XFADE && round == 0 && j != 0) continue;
            bool present = false;
            any = false;
            for (int term = term_lo; term < term_hi; ++term) {
                // descriptor words are workgroup-uniform: keep them in SGPRs
                // through the scalar cache: two short round trips instead of two vector-memory ones in front of the RIR loads
                const i32x4 dw = uniform_load4(d + 4 * term);
                const int ridx = dw.x;
                if (ridx < 0) continue;
                present = true;
                const int L = uniform_load(p.rir_len + ridx);
                const int spec0 = dw.y, m_min = dw.z, m_cnt = dw.w;
                const BankRow br = bank_row(p, ridx, ch);
                const int nbh = (L + kB - 1) / kB;
                for (int i = 0; i < nbh; ++i) {
                    const int m = j - i;
                    if (m < m_min || m >= m_min + m_cnt) continue;
                    // lane id made opaque per iteration: otherwise LICM hoists every lane-invariant address and
                    // predicate of the (large) body out of the loop and the 128-VGPR budget spills them all.
                    int tl = t;
                    SSK_OPAQUE1(tl);
                    if (!any) {
                        conv_block<false, false>(lds, p, tw, tl, br, L, i, spec0 + (m - m_min), acc);
                        any = true;
                    } else {
                        lds_barrier();                   // previous block's item reads of layout B are done
                        conv_block<true, false>(lds, p, tw, tl, br, L, i, spec0 + (m - m_min), acc);
                    }
                }
            }
            if (XFADE && round == 0 && !present) continue;   // no previous RIR (first step of an episode): no blend
            int tl = t;
            if (XFADE) SSK_OPAQUE1(tl);
            if (any) {
                lds_barrier();
                items_to_time(lds, tw, tl, acc, y);
            } else {
#pragma unroll
                for (int a = 0; a < 8; ++a) y[a] = mk2(0.f, 0.f);
            }
            if (XFADE && round == 0) {
                have_prev = true;
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    if (tl + 1024 * a < kPrevPairs) s_prev[tl + 1024 * a] = y[a];
                lds_barrier();                           // pass-1' reads of layout A done before the next pass 1 writes it
            }
        }
        if (XFADE && have_prev) {
            // crossfade(): x1[:, :n+1] * flip(arange(n+1)/n) + x2[:, :n+1] * (arange(n+1)/n), n = fade_len
            const float fl = (float)p.fade_len;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int m = t + 1024 * a, n0 = 2 * m;
                if (n0 <= p.fade_len) {                  // own slot: written by this thread, no barrier needed
                    const c32 yp = s_prev[m];
                    y[a].x = yp.x * ((float)(p.fade_len - n0) / fl) + y[a].x * ((float)n0 / fl);
                    if (n0 + 1 <= p.fade_len)
                        y[a].y = yp.y * ((float)(p.fade_len - n0 - 1) / fl) + y[a].y * ((float)(n0 + 1) / fl);
                }
            }
        }
    }
    int ounit = unit;                                       // where the row's results go (unit table: units are dealt out sorted)
    if constexpr (TAB) ounit = ut.tab[kTabWords * unit + 2];
    if (part == 0) store_row_block(p, t, (size_t)ounit * 2 + ch, j, y);
    if (FUSE) {
        if (t < kNfft) s_win[t] = win_v;                    // visible to the STFT phase after its first barrier
        if (t < 256) s_tw512[posN(t)] = tw512_v;
        fused_stft_phase<WIDE>(lds, p, t, ounit, ch, y, s_win, s_tw512, wq, s_res, part);
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv_spec: the same convolution from a SPECTRAL RIR bank (SURVEY 7 "store the RIR bank as half-spectra":
// twice the bytes, half the flops).  SoundSpaces 1.0 RIRs are static files (simulator.py:615-618), so their block
// spectra H'_i = 2*rFFT_{2kB}(rir[i*kB:(i+1)*kB]) are computed ONCE when the bank is built (ss_rir_spectra_f32:
// k_source_windows run over the bank rows with scale 1) and stored in the register order the item stage consumes,
// exactly like the source-window spectra.  A row is then
//     Y_j = sum_i H'_i * S'_{j-i}   (16 complex multiplies per thread and block pair, straight from two coalesced
//                                    16-byte loads per lane, no LDS)  ->  Hermitian merge + inverse radix-4 -> LDS ->
//     passes 3'-2'-1' -> the row's kB samples in registers  ->  store / fused STFT
// i.e. the forward FFT (passes 1-3, half of the item stage, 4 of the 8 workgroup barriers) is gone.  Per (unit, ear)
// and RIR block the kernel READS 128 KiB of H' (vs 64 KiB of time-domain RIR): the algorithmic bytes of SURVEY 8(d)
// are defined on the time-domain bank, the actual traffic of this kernel is about twice that - reported as such.
// Same unit descriptors as k_conv (term = {bank entry | -1, first window slot, m_min, count}); rir_len[entry] still
// says how many blocks of the entry are non-zero.
__device__ __forceinline__ void spec_block_product(const f32x4* spec, int t, const f32x4* hp, int slot, bool accumulate,
                                                   c32 (&acc)[2][8]) {
    const f32x4* sp = spec + (size_t)slot * (kSpecComplex / 2) + t;
    f32x4 hv[2][4], sv[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            hv[s][hh] = ld_stream(hp + (s * 4 + hh) * 1024);
            sv[s][hh] = sp[(s * 4 + hh) * 1024];
        }
    // ALL sixteen loads are issued before the first multiply: left alone, the scheduler starts the first product after
    // four loads and puts an s_waitcnt vmcnt(2) in front of it (seen in the ISA) - the wave then sits out one full memory
    // latency with a quarter of its loads in flight before it issues the other twelve
    SSK_SCHED_BARRIER();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const c32 h = (e & 1) ? hv[s][e >> 1].zw : hv[s][e >> 1].xy;
            const c32 w = (e & 1) ? sv[s][e >> 1].zw : sv[s][e >> 1].xy;
            c32 pr = cmul(h, w);
            if (s == 0 && e == 0 && t == 0) pr = mk2(h.x * w.x, h.y * w.y);      // (X[0], X[16384]) are real
            if (accumulate) acc[s][e] += pr; else acc[s][e] = pr;
        }
}

template <bool FUSE, bool SIMPLE, bool TAB = false>
__global__ __launch_bounds__(1024) void k_conv_spec(ConvParams p, UnitTab<TAB> ut = UnitTab<TAB>()) {
    static_assert(!TAB || SIMPLE, "the unit table serves the loop-free kernel");
    __shared__ c32 lds[FUSE && 16 * kWaveScratch > kLdsComplex ? 16 * kWaveScratch : kLdsComplex];
    const int t = threadIdx.x;
    // 1-D grid of 2N * nb_y workgroups.  With several output blocks per row (44.1 kHz: 3) the blocks of one row sit NEXT
    // to each other in slot order, i.e. on one XCD at about the same time, so the block spectra H'_i that block j re-reads
    // after block j-1 (i <= j: 6 block reads per row, 3 distinct) come out of that XCD's L2 instead of HBM
    // (44.1 kHz, 128 units: 77.7 -> 67.8 us).
    // every kernel argument the start of the kernel needs, in ONE batch of scalar loads (see SSK_HAVE_S): the chain in
    // front of the row's first vector loads is then arguments -> descriptor, two round trips
    int grid = (int)gridDim.x;
    const f32x4* spec_base = p.spec;
    const f32x4* hspec_base = p.hspec;
    SSK_HAVE_S(grid); SSK_HAVE_S(spec_base); SSK_HAVE_S(hspec_base);
#if defined(SS_LADDER)
    if (p.dbg == 10) return;                            // SS_HIP_DBG=10: the launch alone (dispatch of 2N x 1024 threads)
#endif
    const int slot_all = row_slot(blockIdx.x, grid, p.xcd_map);
    // fused rows (one output block) may be rendered by 2^parts_log2 workgroups each (ConvParams::parts_log2)
    const int part = FUSE ? slot_all & ((1 << p.parts_log2) - 1) : 0, slot = FUSE ? slot_all >> p.parts_log2 : slot_all;
    if (FUSE && part && part * part_blocks(p.t4, p.parts_log2) >= p.t4) return;
    // (the division of two uniform values is done on the vector unit: bring the quotient back to a scalar register)
    const int row = SIMPLE ? slot : __builtin_amdgcn_readfirstlane(slot / p.nb_y), j = SIMPLE ? 0 : slot - row * p.nb_y;
    const int unit = row >> 1, ch = row & 1;
    const int* d = p.desc + 8 * unit;
    __shared__ float s_win[FUSE ? kNfft : 1];
    __shared__ c32 s_tw512[FUSE ? kTw512Lds : 1];
    __shared__ float s_res[FUSE ? kResFloats : 1];          // results of the fused STFT phase (see fused_stft_phase)
    c32 wq = mk2(1.f, 0.f), tw512_v = mk2(0.f, 0.f);
    float win_v = 0.f;
    ThreadTw tw;
    {
        // The thread's table entries (L2 hits, independent of the descriptor) go out FIRST: they travel under the scalar
        // round trips and the row's loads.  Issued after the products (where they are first needed) they were one more
        // exposed round trip in front of the item stage.
        tw = load_thread_tw(p.tb.twM, p.tb.twItem, t);
        if (FUSE) {                                     // unconditional (clamped) loads: a predicated load would be
            win_v = p.tb.win[t & (kNfft - 1)];          // merged with the default value by a register move, i.e. waited
            tw512_v = p.tb.tw512[t & 255];              // for right here; only threads < 512 / < 256 store theirs to LDS
            wq = p.tb.twM[64 * (t & 15)];
        }
    }
    c32 acc[2][8];
    c32 y[8];
    bool any = false;
    const size_t row_f4 = (size_t)p.h_blocks * (kSpecComplex / 2);     // f32x4 per (entry, ear)
    if (SIMPLE) {
        i32x4 dw;
        if constexpr (TAB) dw = i32x4{ut.tab[kTabWords * unit], ut.tab[kTabWords * unit + 1], 0, 1};            // {index | -1, slot of window 0, m_min, count}
        else dw = uniform_load4(d);
        const int ridx = dw.x;
        if (ridx >= 0) {
            // The length word is NOT read here: it would be a third dependent scalar round trip (kernel arguments ->
            // descriptor -> rir_len[ridx]) in front of the row's loads, and all it could say is "empty RIR" - whose
            // block spectrum is exactly zero (rows are zero beyond rir_len, H' = FFT of zeros), so the products and
            // everything after them come out as zeros anyway.
            const f32x4* hp = hspec_base + ((size_t)ridx * 2 + ch) * row_f4 + t;
            if (dw.z <= 0 && dw.z + dw.w > 0) {
                spec_block_product(spec_base, t, hp, dw.y - dw.z, false, acc);
                any = true;
            }
        }
    } else {
        // Scalar round trips in front of the first loads: kernel arguments -> both descriptor terms (one trip).  The
        // length words are read only when a bank entry has several blocks (to skip the all-zero ones); with one block
        // per entry all they could say is "empty RIR", whose block spectrum is exactly zero (see the SIMPLE branch).
        i32x4 dws[2];
        uniform_load8(d, dws[0], dws[1]);
        for (int term = 0; term < 2; ++term) {
            const i32x4 dw = dws[term];
            const int ridx = dw.x;
            if (ridx < 0) continue;
            const int spec0 = dw.y, m_min = dw.z, m_cnt = dw.w;
            const BankSpec bs = bank_spec(p, ridx, ch);
            int nbh = bs.h_blocks;
            if (nbh > 1) nbh = min(nbh, (uniform_load(p.rir_len + ridx) + kB - 1) / kB);
            for (int i = 0; i < nbh; ++i) {
                const int m = j - i;
                if (m < m_min || m >= m_min + m_cnt) continue;
                int tl = t;
                SSK_OPAQUE1(tl);
                const f32x4* hp = bs.hp + (size_t)i * (kSpecComplex / 2) + tl;
                spec_block_product(spec_base, tl, hp, spec0 + (m - m_min), any, acc);
                any = true;
            }
        }
    }
    if (FUSE) {
        // The table values are "arrived" from here on as far as the compiler is concerned (they were the first loads of
        // the kernel; the row's data, fetched after them, has just been consumed).  Left pending, their first use - inside
        // the STFT blocks - is guarded by s_waitcnt vmcnt(0), which (one in-order counter) also waits for whatever was
        // stored just before: the audiogoal row, and in the second block the first block's 65 result stores.
        SSK_OPAQUE2(wq); SSK_OPAQUE2(tw512_v); SSK_OPAQUE1(win_v);
    }
#if defined(SS_LADDER)                                  // early exits of the timing ladder: never in the product build
    if (p.dbg == 1) {                                   // exit after the loads + products
        if (acc[0][0].x == 123.456f && any) p.out[0] = acc[1][7].y + tw.p1.x;
        return;
    }
    if (p.dbg == 2) {                                   // exit after the item stage (Hermitian merge, radix-4, LDS store)
        if (any) { item_store_inv(lds, tw.i0, t, acc[0]); item_store_inv(lds, tw.i1, t + 1024, acc[1]); }
        return;
    }
#endif
    if (any) {
        items_to_time(lds, tw, t, acc, y);
    } else {
#pragma unroll
        for (int a = 0; a < 8; ++a) y[a] = mk2(0.f, 0.f);
    }
#if defined(SS_LADDER)
    if (p.dbg == 3) {                                   // exit before the stores / STFT
        if (y[0].x == 123.456f) p.out[0] = y[7].y;
        return;
    }
#endif
    int ounit = unit;
    if constexpr (TAB) ounit = ut.tab[kTabWords * unit + 2];
    if (part == 0) store_row_block(p, t, (size_t)ounit * 2 + ch, j, y);
    if (FUSE) {
        if (t < kNfft) s_win[t] = win_v;
        if (t < 256) s_tw512[posN(t)] = tw512_v;
        fused_stft_phase(lds, p, t, ounit, ch, y, s_win, s_tw512, wq, s_res, part);
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv_rows: the SIMPLE case as a PERSISTENT kernel for launches with more (unit, ear) rows than CUs.
// One workgroup owns a CU (139 KB of LDS) and walks rows blockIdx.x, + gridDim.x, ...  With one workgroup per row
// (k_conv) a CU's time line is load RIR (HBM latency, nothing to compute) -> 7 FFT passes (no memory traffic) ->
// store, strictly serial because a second workgroup does not fit beside the first (ablation, profiles/r1/NOTES.md:
// 303 us = 203 compute + ~100 exposed IO at 2048 units).  Here the NEXT row's RIR is loaded into 16 registers as
// soon as pass 1 has consumed the current one, so its HBM latency runs under passes 2-3 and the item stage, and the
// current row's stores drain under the next row's pass 1.
// gfx9 has ONE in-order counter (vmcnt) for loads AND stores, so the order of issue is part of the design:
//   * the row descriptors come through the scalar cache (s_load, lgkmcnt): a vector load here would have to wait
//     for the previous row's freshly issued stores;
//   * prefetch loads are issued before the window-spectrum loads; the wait for the latter (vmcnt(0) at the second
//     item, long after) therefore retires them too and nothing is pending on the RIR registers at the back edge;
//   * the prologue load is waited for explicitly before the loop (otherwise the merged loop-header state would make
//     the compiler wait at the top of EVERY iteration, i.e. for the stores just issued).
// Preconditions checked by the launcher: planar bank rows (elem stride 1), even capacity <= kB, 8-byte aligned rows.
struct RowInfo { int active, slot; const c32* h2; };



__device__ __forceinline__ RowInfo row_info(const ConvParams& p, int row) {
    RowInfo r{0, 0, reinterpret_cast<const c32*>(p.rir)};      // inactive: a valid address for the dummy prefetch
    const i32x4 d = uniform_load4(p.desc + 8 * (row >> 1));    // {rir index, first slot, m_min, count}: one round trip
    const int ridx = d.x;
    if (ridx < 0) return r;
    const int L = uniform_load(p.rir_len + ridx);              // the only dependent access
    if (L > 0 && d.z <= 0 && d.z + d.w > 0) {
        r.active = 1;
        r.slot = d.y - d.z;
        r.h2 = reinterpret_cast<const c32*>(p.rir + (size_t)ridx * p.rir_unit_stride + (size_t)(row & 1) * p.rir_chan_stride);
    }
    return r;
}

// Issue the loads of the row's packed sample pairs t + 1024 a, a < 8, WITHOUT waiting for them.  Inline asm on
// purpose: (1) the compiler turns the clamped-address form back into eight predicated branches, and (2) it guards the
// destination registers with s_waitcnt vmcnt(0), which on gfx9 also waits for the previous row's freshly issued
// stores.  The registers must not be touched until row_rir_wait(); out-of-row lanes load pair 0 and are zeroed there.
__device__ __forceinline__ void row_rir_issue(const ConvParams& p, const RowInfo& r, int t, c32 (&h)[8]) {
    const int m_end = p.rir_cap >> 1;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int m = t + 1024 * a;
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned off = m < m_end ? 8u * (unsigned)m : 0u;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(h[a]) : "v"(off), "s"(r.h2) : "memory");
#else
        h[a] = r.h2[m < m_end ? m : 0];
#endif
    }
}

__device__ __forceinline__ void row_rir_wait(const ConvParams& p, int active, int t, c32 (&h)[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
    // every VMEM operation of this wave retired.  The wait carries no register operands: tied operands may be
    // realised as copies placed BEFORE the statement, i.e. reads of registers whose loads are still in flight.  The
    // second statement makes the compiler treat the registers as produced here (no use can move above it).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]));
#endif
    const int m_end = active ? p.rir_cap >> 1 : 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) h[a] = (t + 1024 * a < m_end) ? h[a] : mk2(0.f, 0.f);
}

// Conv-only: for the fused kernel the same scheme was measured and rejected (profiles/r1/NOTES.md) - its STFT phase
// has no 16 registers to spare (20 spills, +7 %), and an L2-only prefetch (dummy loads, real loads at the top of the
// next row) gained nothing (-1 % at 2048 units, +3 % at 512).
__global__ __launch_bounds__(1024) void k_conv_rows(ConvParams p, int n_rows) {
    __shared__ c32 lds[kLdsComplex];
    const int t = threadIdx.x;
    ThreadTw tw = load_thread_tw(p.tb.twM, p.tb.twItem, t);
    int row = blockIdx.x;
    RowInfo cur = row_info(p, row);
    c32 h[8];
    row_rir_issue(p, cur, t, h);                            // unconditional, like the one in the loop
    row_rir_wait(p, cur.active, t, h);
    // nothing the compiler tracks may still be pending when the loop starts: its merged loop-header state would
    // otherwise put a partial s_waitcnt vmcnt(n) for these table loads inside the body, where it would (the count
    // being in-order) also wait for the prefetch loads the compiler does not know about
    SSK_OPAQUE2(tw.p1); SSK_OPAQUE2(tw.p2); SSK_OPAQUE2(tw.i0); SSK_OPAQUE2(tw.i1);
    for (;;) {
        int tl = t;
        SSK_OPAQUE1(tl);                                    // see k_conv: keeps LICM from hoisting the body's addresses
        const int nxt = row + (int)gridDim.x;
        if (cur.active) pass1_fwd<true>(lds, tw.p1, tl, [&](int m) { return h[(m - tl) >> 10]; });
        RowInfo nx{0, 0, reinterpret_cast<const c32*>(p.rir)};
        if (nxt < n_rows) nx = row_info(p, nxt);
        if (cur.active) simple_row_fwd(lds, p, tw, tl, cur.slot);
        // The prefetch goes out AFTER the last window-spectrum load of this row has been waited for: the counter is
        // in-order, so HBM loads issued earlier would sit in front of those L2 hits and every partial
        // s_waitcnt vmcnt(n) the compiler emits for them (n computed without these loads) would wait for the prefetch
        // (measured: issued right after pass 1, -3 %; here, -8 %).  From here it has the three inverse passes
        // (~4 us) to arrive.  ONE issue site, executed unconditionally (an inactive next row reads pair 0 of the bank
        // and is zeroed at the wait): two sites would meet in a register merge, i.e. copies of registers whose loads
        // are still in flight.
        row_rir_issue(p, nx, tl, h);
        c32 y[8];
        if (cur.active) simple_row_inv(lds, tw, tl, y);
        else {
#pragma unroll
            for (int a = 0; a < 8; ++a) y[a] = mk2(0.f, 0.f);
        }
        // retire the prefetch BEFORE this row's stores are issued: one in-order counter serves loads and stores
        row_rir_wait(p, nx.active, tl, h);
        store_row_block(p, tl, (size_t)row, 0, y);
        if (nxt >= n_rows) break;
        row = nxt;
        cur = nx;
        lds_barrier();                                      // every wave is done with the LDS buffer of this row
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv_spec_rows: the loop-free spectral case (AudioGoal only) as a PERSISTENT kernel for launches with more rows than
// CUs.  k_conv_spec's time line on a CU is: H' streams in (128 KiB at the ~26 GB/s a CU ingests from HBM = 5 us, nothing to
// compute) -> products + 3 inverse passes (~6 us of instruction issue, no memory traffic) -> stores.  Here the NEXT row's
// H' is loaded into the 32 registers the products have just freed, so it streams in under the inverse passes; the window
// spectrum of a row (L2 hits) is loaded at the row's start, and a workgroup takes both ears of a unit back to back, so the
// second ear finds that spectrum in the L2 it has just been pulled through.
// Same in-order vmcnt discipline as k_conv_rows: descriptors through the scalar cache, ONE unconditional asm issue site
// for the prefetch, placed after the row's last compiler-visible load has been waited for, retired before the row's
// stores are issued (build-time ISA guard: scripts/check_prefetch_regs.py).
struct SpecRowInfo { int active, slot; const f32x4* hp; };

__device__ __forceinline__ SpecRowInfo spec_row_info(const ConvParams& p, int row) {
    SpecRowInfo r{0, 0, p.hspec};                              // inactive: a valid address for the dummy prefetch
    const i32x4 d = uniform_load4(p.desc + 8 * (row >> 1));
    const int ridx = d.x;
    if (ridx < 0) return r;
    if (d.z <= 0 && d.z + d.w > 0) {                          // (an empty RIR's H' is zero: see k_conv_spec)
        r.active = 1;
        r.slot = d.y - d.z;
        r.hp = p.hspec + ((size_t)ridx * 2 + (row & 1)) * (size_t)p.h_blocks * (kSpecComplex / 2);
    }
    return r;
}

// (item 0's half of H': the whole row, 32 registers in flight under the inverse passes, spilled at the 128-VGPR cap)
__device__ __forceinline__ void spec_row_issue(const SpecRowInfo& r, int t, f32x4 (&h)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned off = 16u * (unsigned)(t + 1024 * k);
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(h[k]) : "v"(off), "s"(r.hp) : "memory");
#else
        h[k] = r.hp[t + 1024 * k];
#endif
    }
}

__device__ __forceinline__ void spec_row_wait(f32x4 (&h)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
#endif
}

__global__ __launch_bounds__(1024) void k_conv_spec_rows(ConvParams p, int n_rows) {
    __shared__ c32 lds[kLdsComplex];
    const int t = threadIdx.x;
    ThreadTw tw = load_thread_tw(p.tb.twM, p.tb.twItem, t);
    // rows are walked in (unit, ear) pairs: workgroup b takes units b, b + G, ...; row = 2 * unit + ear
    int unit = blockIdx.x, ear = 0;
    SpecRowInfo cur = spec_row_info(p, 2 * unit);
    f32x4 h[4];
    spec_row_issue(cur, t, h);
    spec_row_wait(h);
    SSK_OPAQUE2(tw.p1); SSK_OPAQUE2(tw.p2); SSK_OPAQUE2(tw.i0); SSK_OPAQUE2(tw.i1);   // see k_conv_rows
    for (;;) {
        int tl = t;
        SSK_OPAQUE1(tl);
        const int row = 2 * unit + ear;
        const int n_unit = ear ? unit + (int)gridDim.x : unit, n_ear = ear ^ 1, nxt = 2 * n_unit + n_ear;
        SpecRowInfo nx{0, 0, p.hspec};
        if (nxt < n_rows) nx = spec_row_info(p, nxt);
        // item by item (load -> product -> Hermitian merge + inverse radix-4 -> LDS), so that no 32-register accumulator
        // is alive next to H' and the window spectrum.  Item 0's half of H' was prefetched; item 1's half and the window
        // spectrum (L2 hits) are loaded here, item 1's under item 0's arithmetic.
        c32 v1[8];
        if (cur.active) {
            const f32x4* sp = p.spec + (size_t)cur.slot * (kSpecComplex / 2) + tl;
            f32x4 sv[4], h1[4], sw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = sp[k * 1024];
#pragma unroll
            for (int k = 0; k < 4; ++k) { h1[k] = ld_stream(cur.hp + tl + (4 + k) * 1024); sw[k] = sp[(4 + k) * 1024]; }
            c32 v0[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f32x4 hq = h[e >> 1], sq = sv[e >> 1];
                const c32 hh = (e & 1) ? hq.zw : hq.xy, w = (e & 1) ? sq.zw : sq.xy;
                v0[e] = (e == 0 && tl == 0) ? mk2(hh.x * w.x, hh.y * w.y) : cmul(hh, w);    // (X[0], X[16384]) are real
            }
            item_store_inv(lds, tw.i0, tl, v0);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f32x4 hq = h1[e >> 1], sq = sw[e >> 1];
                v1[e] = cmul((e & 1) ? hq.zw : hq.xy, (e & 1) ? sq.zw : sq.xy);
            }
        }
        // every compiler-visible load of this row has been consumed (item 1's products exist): the prefetch of the next
        // row's H' goes out now, into registers that are dead until the next iteration.  ONE unconditional issue site.
        spec_row_issue(nx, tl, h);
        if (cur.active) item_store_inv(lds, tw.i1, tl + 1024, v1);
        c32 y[8];
        if (cur.active) simple_row_inv(lds, tw, tl, y);
        else {
#pragma unroll
            for (int a = 0; a < 8; ++a) y[a] = mk2(0.f, 0.f);
        }
        spec_row_wait(h);                                   // retired BEFORE this row's stores are issued (one in-order counter)
        store_row_block(p, tl, (size_t)row, 0, y);
        if (nxt >= n_rows) break;
        unit = n_unit;
        ear = n_ear;
        cur = nx;
        lds_barrier();                                      // every wave is done with the LDS buffer of this row
    }
}

// ---------------------------------------------------------------------------------------------
// k_obs_rows: the FUSED observation for rows longer than one partition block - 44.1 kHz, the reference's own Replica
// rate (configs/audionav/av_nav/replica/audiogoal.yaml:18): simulator.py:629-632 on 44100-sample rows followed by
// nav.py:86-100 -> (65, 69, 2).  One launch; the waveform never leaves the CU unless the caller wants the audiogoal.
//
// A workgroup owns one CU (the LDS allows one) and walks (unit, ear) rows: row_slot(b) + k * gridDim.x.  Per row it
// renders the output blocks j = 0 .. nb_rows-1 IN ORDER and streams the STFT behind them:
//
//   convolution of block j:  Y_j = sum over the PAIRS (term, RIR block i) with a stored window m = j - i of
//     H'_i * S'_{j-i} (as k_conv / k_conv_spec).  A block spectrum H'_i is either in memory - the spectral bank, or the
//     workgroup's STASH in global memory, where the time-domain path leaves every spectrum it computes that is needed
//     again (kernels' register order: the on-the-fly equivalent of the spectral bank, L2 / Infinity-Cache resident) -
//     or NEW: its forward FFT has not run in this row yet.  Every RIR block is transformed ONCE per row (k_conv re-ran
//     the forward FFTs per output block: 6 instead of 3 at 44.1 kHz).  A block is then
//        1. for every new pair but the last: forward FFT -> item stage -> stash            (multi-second clips at j = 0,
//           distractor terms; never for a 1-s clip, whose block j has exactly one new pair, RIR block j);
//        2. ONE item stage, item by item and IN PLACE (as in k_conv<SIMPLE>):  v = H'_new from LDS (the last new pair's
//           forward FFT, if any) -> stash it if a later block needs it -> v *= S'[its window] -> v += sum over the pairs
//           in memory of H' * S' (two coalesced 16-byte loads per lane and product) -> Hermitian merge + inverse radix-4
//           -> back into the same LDS slots;
//        3. inverse passes 3'-2'-1' -> the block's kB samples in registers.
//     No accumulator exists outside the item stage (the loop kernel carries 32 VGPRs across its forward passes), which is
//     what lets the first window spectrum of item 0 travel under pass 3.
//   STFT behind block j:  hann(400) centred in 512 with hop 160: pooled time block b (frames 4b .. 4b+3) needs samples
//     [640 b - 256, 640 b + 736).  After block j the row is known up to (j+1) kB, so the pooled blocks
//     b0_j <= b < b1_j = floor((floor(((j+1) kB - 256) / 160) + 1) / 4) are complete (44.1 kHz: 25 + 26 + 18 = 69).
//     The block's samples go from registers into LDS behind a CONTEXT of the samples [640 b0_j - 256, j kB) that the
//     previous block left in s_tail (<= 640 floats; block 0: librosa's left centre padding), the last block appends the
//     right padding, and the pooled blocks run exactly as in the one-block kernels (fused_stft_phase): two rounds of
//     16 waves, wave-private scratch overlaying the row buffer.
// Unit descriptors, silent units, distractor term, n_valid < out_len (0.25-s SS2.0 steps: blocks beyond n_valid are
// zeros without any transform) as in k_conv.  SS_FLAG_CROSSFADE (XFADE instantiation, time-domain bank; continuous_simulator.py:47-53,
// 422-424): term 1 is the previous step's RIR and only shapes samples 0..fade_len, all in block 0 - block 0 is rendered
// twice, first from term 1's pairs alone (its head parks in LDS, in the space s_res / s_tail do not use before the
// block's STFT phase), then from term 0's, and blended exactly as in k_conv<XFADE>; later blocks see term 0 only.
// timing ablations (scripts/gpu_rows_ladder.sh builds with -DSS_ROWS_ABL=<mask>; results are WRONG, only the time is read):
//   1 no STFT phase   2 no products from memory   4 no window-spectrum loads   8 no forward passes 2-3   16 no inverse passes
#if defined(SS_ROWS_ABL)
constexpr int kRowsAbl = SS_ROWS_ABL;
#else
constexpr int kRowsAbl = 0;
#endif
constexpr int kTailFloats = 640;            // context handed from output block j to j+1 (nb_rows <= 3: 640, 384)
constexpr int kRowsMaxBlocks = 27;          // pooled blocks behind one output block (32768-sample rows: 25 + 27)
constexpr int kRowsResFloats = kBins4 * kRowsMaxBlocks;
constexpr int kRowsMaxNbh = 16;             // RIR blocks per term the pair masks can hold (2 terms x 16 bits)

// pooled time blocks that are complete once the row is known up to sample `known` (not the row's end)
__host__ __device__ constexpr int pooled_blocks_complete(int known) {
    return (known < kNfft / 2) ? 0 : (((known - kNfft / 2) / kHop + 1) / kPool);
}

// forward passes 1-3 of RIR block i of a time-domain bank row (the item stage follows at the caller)
__device__ __forceinline__ void rows_forward(c32* lds, const ThreadTw& tw, int t, const BankRow& br, int i) {
    const float* h = br.h;
    const int lo = i * kB, es = br.es, cap = br.cap;
    if (es == 1 && !(cap & 1) && !(reinterpret_cast<size_t>(h) & 7)) {       // planar, 8-byte aligned rows
        const c32* h2 = reinterpret_cast<const c32*>(h + lo);
        const int m_end = (cap - lo) >> 1;
        pass1_fwd<true>(lds, tw.p1, t, [&](int m) { return m < m_end ? ld_stream(h2 + m) : mk2(0.f, 0.f); });
    } else {
        pass1_fwd<true>(lds, tw.p1, t, [&](int m) {
            const int n = lo + 2 * m;
            return mk2(n < cap ? h[(size_t)n * es] : 0.f, n + 1 < cap ? h[(size_t)(n + 1) * es] : 0.f);
        });
    }
    lds_barrier();
    pass2<false>(lds, tw.p2, t);
    lds_barrier();
}

// v += H'[item s] * S'[item s] for one pair in memory (hp / sp: per-thread pointers of the two block spectra)
__device__ __forceinline__ void rows_product(const f32x4* hp, const f32x4* sp, int t, int s, c32 (&v)[8]) {
    f32x4 hv[4], sv[4];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) { hv[hh] = ld_stream(hp + (s * 4 + hh) * 1024); sv[hh] = sp[(s * 4 + hh) * 1024]; }
    if (kRowsAbl & 4) {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) sv[hh] = f32x4{1.f, 0.f, 1.f, 0.f};
    }
    SSK_SCHED_BARRIER();                                  // all eight loads before the first multiply (see spec_block_product)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const c32 h = (e & 1) ? hv[e >> 1].zw : hv[e >> 1].xy, w = (e & 1) ? sv[e >> 1].zw : sv[e >> 1].xy;
        c32 pr = cmul(h, w);
        if (s == 0 && e == 0 && t == 0) pr = mk2(h.x * w.x, h.y * w.y);      // (X[0], X[16384]) are real
        v[e] += pr;
    }
}

// Hand-off between WORKGROUPS of one launch (k_obs_blocks: the samples an output block leaves for the STFT frames that straddle
// into the next block travel through global memory): agent-scope release / acquire on a flag word, the data itself read with
// agent-scope loads (the XCDs' L2s are not coherent with each other for plain accesses).  The wait is BOUNDED: a consumer
// that never sees its flag goes on with whatever the buffer holds - wrong numbers that the parity tests catch, never a hung GPU.
struct BlockSync { int* flag_out; const int* flag_in; int epoch; };
__device__ __forceinline__ void flag_release(int* f, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
    *f = v;
#endif
}
__device__ __forceinline__ bool flag_acquire(const int* f, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int spin = 0; spin < (1 << 20); ++spin) {
        if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == v) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
#else
    return *f == v;                                       // (host build: workgroups run one after the other, producers first)
#endif
}
__device__ __forceinline__ void wait_global_stores() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) alone (expcnt 7, lgkmcnt 15: untouched): this wave's loads AND stores
#endif
}
__device__ __forceinline__ float ld_agent(const float* q) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *q;
#endif
}

// L2 warm-up: one 4-byte load per 128-byte line of [ptr, ptr + bytes), fire and forget - the data goes to a 256-byte dummy in LDS
// (global_load_lds: no destination register, nothing ever waits for it), the LINES stay in this XCD's L2.  Issued in front of an
// STFT phase (microseconds of pure LDS / VALU work) for what the NEXT output block will load cold from HBM: its RIR block or
// (k_obs_rows; -DSS_ROWS_NO_L2_WARM is the A/B switch)
__device__ __forceinline__ void l2_touch(const void* ptr, int bytes, int t, int* s_dummy) {
#if defined(__HIP_DEVICE_COMPILE__)
    const char* c = static_cast<const char*>(ptr);
    for (int off = t * 128; off < bytes; off += kT * 128)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const int*>(c + off), s_dummy, 4, 0, 0);
#else
    (void)ptr; (void)bytes; (void)t; (void)s_dummy;
#endif
}

// STFT of the pooled blocks [b0, b1) that output block j completes (see the kernel comment).  y = the block's kB samples
// (packed pairs t + 1024 a).  `last`: j is the row's last block (right centre padding, frames up to n_frames - 1).
// tail_in: the previous block's last samples (k_obs_rows: the workgroup's own LDS; GLOBAL_TAIL, k_obs_blocks: global memory
// written by the workgroup that rendered block j - 1, waited for through sync.flag_in); tail_out (may be null): where the next
// block's context goes, sync.flag_out (GLOBAL_TAIL) is released behind it.
template <bool GLOBAL_TAIL = false>
__device__ __forceinline__ void rows_stft_phase(c32* lds, const ConvParams& p, int t, int unit, int ch, int j, int b0, int b1,
                                                bool last, const c32 (&y)[8], const float* s_win, const c32* s_tw512,
                                                const c32* s_wq, float* s_res, const float* tail_in, float* tail_out,
                                                int part = 0, BlockSync sync = BlockSync{nullptr, nullptr, 0}) {
    float* buf = reinterpret_cast<float*>(lds);           // buf[k] = row sample 640 b0 - 256 + k
    const int base = kB * j;                              // first sample of this block
    const int ctx = j == 0 ? kNfft / 2 : base - (kHop * kPool * b0 - kNfft / 2);    // even, <= kTailFloats
    float* yl = buf + ctx;                                // sample `base`
    const int len = p.out_len;
    lds_barrier();                                        // every earlier use of the buffer (pass 1', previous STFT) is over
    bool tail_ok = true;
    {
        c32* yl2 = reinterpret_cast<c32*>(yl) + t;        // one ds_write_b64 per packed pair
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int n = base + 2 * (t + 1024 * a);
            yl2[1024 * a] = mk2(n < p.n_valid ? y[a].x : 0.f, n + 1 < p.n_valid ? y[a].y : 0.f);   // zeros beyond n_valid
        }
        if (GLOBAL_TAIL) {
            if (j > 0 && t == 0) tail_ok = flag_acquire(sync.flag_in, sync.epoch); // block j - 1 has left its last samples
        } else if (j > 0 && t < ctx) buf[t] = tail_in[t]; // the previous block's last samples
    }
    lds_barrier();
    // (a wait that ran out - the producer never came: not reachable under the launcher's "grid fits the chip" rule - poisons the
    //  block's first frames instead of rendering them from stale samples: NaN in the observation, not a plausible wrong value)
    if (GLOBAL_TAIL && j > 0 && t < ctx) buf[t] = tail_ok ? ld_agent(tail_in + t) : __builtin_nanf("");
    if (j == 0 && t < kNfft / 2) yl[-1 - t] = p.pad_mode == 0 ? yl[1 + t] : 0.f;              // left centre padding
    if (last && t >= 256 && t < 256 + kNfft / 2) {        // right centre padding: sample len + k = sample len - 2 - k
        const int k = t - 256;
        yl[len - base + k] = p.pad_mode == 0 ? yl[len - base - 2 - k] : 0.f;
    }
    if (!last && tail_out) {                              // context of the next block: samples [640 b1 - 256, base + kB)
        const int s0n = kHop * kPool * b1 - kNfft / 2;
        if (t < kTailFloats && s0n + t < base + kB) tail_out[t] = yl[s0n - base + t];
        // GLOBAL_TAIL: the stores of EVERY thread must have reached L2 before thread 0 releases the flag - lds_barrier() only
        // waits for LDS operations (lgkmcnt), and the agent-scope release below only for thread 0's own wave
        if (GLOBAL_TAIL) wait_global_stores();
    }
    lds_barrier();
    if (GLOBAL_TAIL && sync.flag_out && t == 0) flag_release(sync.flag_out, sync.epoch);      // (behind the barrier: every
                                                          //  thread's tail store happened-before this agent-scope release)
    const int lane = t & 63, wv = t >> 6, cnt = b1 - b0;
    const c32 wq = s_wq[lane & 15];
    // frames relative to the buffer: pooled block b0 + k starts at buf + 640 k; both rounds are pulled into registers
    // before the wave scratches overlay the buffer (see fused_stft_phase)
    c32 x0[16], x1[16];
    // split rows (ConvParams::parts_log2: small steps, one row on 2 / 4 / 8 CUs): every part has rendered the whole block;
    // of the phase's pooled blocks it computes [k_lo, k_hi) only
    const int per = (cnt + (1 << p.parts_log2) - 1) >> p.parts_log2, k_lo = part * per, k_hi = min(cnt, k_lo + per);
    const int kw = k_lo + wv;
    const bool one = kw < k_hi, two = kw + 16 < k_hi;
    const int live = p.n_frames - kPool * b0;             // frames of this phase that exist (relative index < live)
    stft_load_padded(buf, 4 * kw + (lane >> 4), one ? live : 0, lane & 15, s_win, x0);      // (the window pairs in registers for
    stft_load_padded(buf, 4 * (kw + 16) + (lane >> 4), two ? live : 0, lane & 15, s_win, x1);  // both rounds spill here: 8-16 VGPRs)
    lds_barrier();
    if (one) stft_block(lds + wv * kWaveScratch, lane, wq, s_tw512, x0, [&](int b, float v) { s_res[b * cnt + kw] = v; });
    if (two) {
        wave_sync();
        stft_block(lds + wv * kWaveScratch, lane, wq, s_tw512, x1, [&](int b, float v) { s_res[b * cnt + kw + 16] = v; });
    }
    lds_barrier();
    float* o = p.sgram + ((size_t)unit * kBins4 * p.t4 + b0) * 2 + ch;
    for (int b = t / 32; b < kBins4; b += kT / 32) {      // 32 threads per pooled row (cnt <= 27): no division by cnt
        const int k = t & 31;
        if (k >= k_lo && k < k_hi) o[((size_t)b * p.t4 + k) * 2] = s_res[b * cnt + k];
    }
}

template <bool SPECTRAL, bool XFADE = false, bool BUCKETS = true>
__global__ __launch_bounds__(1024) void k_obs_rows(ConvParams p, int n_rows) {
    static_assert(!(SPECTRAL && XFADE), "cross-faded rows are rendered from the time-domain bank");
    __shared__ c32 lds[16 * kWaveScratch > kLdsComplex ? 16 * kWaveScratch : kLdsComplex];
    __shared__ float s_win[kNfft];
    __shared__ c32 s_tw512[kTw512Lds];
    // pooled results of an STFT phase | context handed from block j to j+1; before block 0's STFT phase both are free and
    // hold the head of the cross-faded row convolved with the previous RIR (kPrevPairs packed pairs)
    constexpr int kRtFloats = XFADE && 2 * kPrevPairs > kRowsResFloats + kTailFloats + 1 ? 2 * kPrevPairs
                                                                                       : kRowsResFloats + kTailFloats + 1;
    __shared__ __attribute__((aligned(8))) float s_rt[kRtFloats];
    static_assert(!XFADE || kRtFloats * sizeof(float) >= kPrevPairs * sizeof(c32), "s_prev overlay");
    float* s_res = s_rt;
    float* s_tail = s_rt + kRowsResFloats + (kRowsResFloats & 1);        // (even offset: 8-byte aligned packed reads)
    c32* s_prev = reinterpret_cast<c32*>(s_rt);
    const int t = threadIdx.x;
    ThreadTw tw = load_thread_tw(p.tb.twM, p.tb.twItem, t);
    {   // the STFT's tables, staged once per workgroup (unconditional clamped loads: see k_conv)
        const float win_v = p.tb.win[t & (kNfft - 1)];
        const c32 tw512_v = p.tb.tw512[t & 255];
        if (t < kNfft) s_win[t] = win_v;
        if (t < 256) s_tw512[posN(t)] = tw512_v;
    }
    // the STFT's 256-point twiddle exp(-2 pi i q / 256), q = lane & 15: parked in LDS, read back at the start of every STFT
    // phase (two registers that would otherwise be live across the whole row loop of a kernel at the 128-VGPR limit)
    __shared__ c32 s_wq[16];
    {
        const c32 wq0 = p.tb.twM[64 * (t & 15)];
        if (t < 16) s_wq[t] = wq0;
    }
    __shared__ int s_dummy[64];                           // sink of the L2 warm-up loads (l2_touch)
    // nothing the compiler tracks may be pending when the row loop starts (see k_conv_rows)
    SSK_OPAQUE2(tw.p1); SSK_OPAQUE2(tw.p2); SSK_OPAQUE2(tw.i0); SSK_OPAQUE2(tw.i1);
    const int G = (int)gridDim.x;
    const int nb_rows = (p.out_len + kB - 1) / kB;
    const size_t blk_f4 = kSpecComplex / 2;               // f32x4 per block spectrum
    // the workgroup's stash: [term][RIR block] block spectra of the row being rendered (time-domain bank only)
    f32x4* stash = SPECTRAL ? nullptr : p.stash + (size_t)blockIdx.x * p.stash_terms * p.stash_nbh * blk_f4;
    // split rows (parts_log2 > 0; the launcher only asks for it when every (row, part) has a workgroup of its own): slot =
    // (row, part); every part renders the row's blocks, the pooled STFT blocks of each phase are shared out (rows_stft_phase)
    const int part = row_slot(blockIdx.x, G, p.xcd_map) & ((1 << p.parts_log2) - 1);
    for (int row = row_slot(blockIdx.x, G, p.xcd_map) >> p.parts_log2; row < n_rows; row += G) {
        const int unit = row >> 1, ch = row & 1;
        i32x4 dws[2];
        uniform_load8(p.desc + 8 * unit, dws[0], dws[1]);
        if (p.n_terms < 2) dws[1].x = -1;                 // SS_FLAG_NO_DISTRACTOR: term 1 is ignored, as in k_conv<SIMPLE>
        // (scalars, not arrays indexed by `term`: a dynamically indexed array lives in scratch memory)
        int nbh0 = 0, nbh1 = 0;
        if (dws[0].x >= 0) nbh0 = min(kRowsMaxNbh, (uniform_load(p.rir_len + dws[0].x) + kB - 1) / kB);
        if (dws[1].x >= 0) nbh1 = min(kRowsMaxNbh, (uniform_load(p.rir_len + dws[1].x) + kB - 1) / kB);
        if (SPECTRAL) {
            if (dws[0].x >= 0) nbh0 = min(nbh0, bank_spec<BUCKETS>(p, dws[0].x, 0).h_blocks);
            if (dws[1].x >= 0) nbh1 = min(nbh1, bank_spec<BUCKETS>(p, dws[1].x, 0).h_blocks);
        } else {                                          // (never more blocks than the entry's row - and the stash - holds)
            if (dws[0].x >= 0) nbh0 = min(nbh0, min(p.stash_nbh, (bank_row<BUCKETS>(p, dws[0].x, 0).cap + kB - 1) / kB));
            if (dws[1].x >= 0) nbh1 = min(nbh1, min(p.stash_nbh, (bank_row<BUCKETS>(p, dws[1].x, 0).cap + kB - 1) / kB));
        }
        if (dws[0].x < 0 && dws[1].x < 0) {               // silent unit (simulator.py:610-612): exact zeros, no transforms
            if (part) continue;
            int tz = t;
            SSK_OPAQUE1(tz);                              // nothing of these loops is worth a register outside them
            if (p.out) {
#pragma nounroll
                for (int n = tz; n < p.out_len; n += kT) p.out[(size_t)row * p.out_len + n] = 0.f;
            }
            float* o = p.sgram + (size_t)unit * kBins4 * p.t4 * 2 + ch;
#pragma nounroll
            for (int e = tz; e < kBins4 * p.t4; e += kT) o[2 * e] = 0.f;
            continue;
        }
        // pair p = term * 16 + i.  in_mem: H'_i of the term can be loaded (spectral bank: always; else: it is in the stash)
        unsigned in_mem = SPECTRAL ? 0xffffffffu : 0u;
        int b0 = 0;
        for (int j = 0; j < nb_rows; ++j) {
            int tl = t;
            SSK_OPAQUE1(tl);                              // see k_conv: keeps LICM from hoisting the body's addresses
            c32 y[8];
            // ---- which pairs does block j use?  (wave-uniform bit masks)
            unsigned want = 0;
            if (j < p.nb_y) {
#pragma unroll
                for (int term = 0; term < 2; ++term) {
                    const i32x4 dw = term ? dws[1] : dws[0];
                    const int nb = term ? nbh1 : nbh0;
                    if (dw.x < 0) continue;
                    // m = j - i in [m_min, m_min + cnt)  <=>  i in (j - m_min - cnt, j - m_min]
                    const int i_hi = min(nb - 1, j - dw.z), i_lo = max(0, j - dw.z - dw.w + 1);
                    if (i_hi >= i_lo) want |= (((2u << i_hi) - 1u) & ~((1u << i_lo) - 1u)) << (16 * term);
                }
            }
            // cross-fade: block 0 twice - round 0 from the previous RIR's pairs (term 1), round 1 from the current one's;
            // the previous RIR plays no part in later blocks
            const bool xf = XFADE && dws[1].x >= 0;       // (an instantiation of its own: the plain rows pay nothing for it)
            const unsigned want_all = (xf && j > 0) ? (want & 0xffffu) : want;
            for (int round = (xf && j == 0) ? 0 : 1; round < 2; ++round) {
            want = (xf && j == 0) ? (round == 0 ? (want_all & 0xffff0000u) : (want_all & 0xffffu)) : want_all;
            unsigned fresh = want & ~in_mem;              // pairs whose forward FFT has to run now
            if (want) {
                // ---- 1. every new pair but the last: forward FFT -> stash
                int last_new = -1;                        // the pair whose item stage carries the products
                while (fresh) {
                    const int pr = __builtin_ctz(fresh);
                    fresh &= fresh - 1;
                    const int term = pr >> 4, i = pr & 15;
                    int ti = tl;
                    SSK_OPAQUE1(ti);                      // per transform: see k_conv
                    const BankRow br = bank_row<BUCKETS>(p, term ? dws[1].x : dws[0].x, ch);
                    if (last_new >= 0 || b0 > 0 || j > 0) lds_barrier();      // (the LDS buffer's previous readers are done)
                    rows_forward(lds, tw, ti, br, i);
                    if (fresh) {                          // not the last one: its spectrum just goes to the stash
                        pass3_fwd(lds, ti);
                        lds_barrier();
                        f32x4* st = stash + (size_t)(term * p.stash_nbh + i) * blk_f4 + ti;
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            c32 v[8];
                            item_load_fwd(lds, s ? tw.i1 : tw.i0, ti + 1024 * s, v);
#pragma unroll
                            for (int hh = 0; hh < 4; ++hh) st[(s * 4 + hh) * 1024] = mk4(v[2 * hh], v[2 * hh + 1]);
                        }
                        in_mem |= 1u << pr;
                    }
                    last_new = pr;
                }
                // ---- 2. the item stage of the block
                int ti = tl;
                SSK_OPAQUE1(ti);
                const bool has_new = last_new >= 0;
                const int n_term = last_new >> 4, n_i = last_new & 15;
                const i32x4 ndw = n_term ? dws[1] : dws[0];
                // the new pair's window spectrum, item 0: under pass 3
                const f32x4* spn = p.spec + (size_t)(ndw.y + (j - n_i - ndw.z)) * blk_f4 + ti;
                f32x4 sn[4];
                if (has_new) {
#pragma unroll
                    for (int hh = 0; hh < 4; ++hh) sn[hh] = spn[hh * 1024];
                    if (!(kRowsAbl & 8)) pass3_fwd(lds, ti);
                    lds_barrier();
                }
                // needed by a later output block of this row?  (m = j - i grows with j)
                const bool keep = has_new && !SPECTRAL && j + 1 < p.nb_y && (j - n_i) + 1 < ndw.z + ndw.w;
                const unsigned mem_pairs = (kRowsAbl & 2) ? 0u : (want & in_mem);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    c32 v[8];
                    if (has_new) {
                        if (s == 1) {
#pragma unroll
                            for (int hh = 0; hh < 4; ++hh) sn[hh] = spn[(4 + hh) * 1024];
                        }
                        item_load_fwd(lds, s ? tw.i1 : tw.i0, ti + 1024 * s, v);
                        if (keep) {
                            f32x4* st = stash + (size_t)(n_term * p.stash_nbh + n_i) * blk_f4 + ti;
#pragma unroll
                            for (int hh = 0; hh < 4; ++hh) st[(s * 4 + hh) * 1024] = mk4(v[2 * hh], v[2 * hh + 1]);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const c32 w = (kRowsAbl & 4) ? mk2(1.f, 0.f) : ((e & 1) ? sn[e >> 1].zw : sn[e >> 1].xy);
                            c32 pr = cmul(v[e], w);
                            if (s == 0 && e == 0 && ti == 0) pr = mk2(v[0].x * w.x, v[0].y * w.y);   // (X[0], X[16384]) are real
                            v[e] = pr;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = mk2(0.f, 0.f);
                    }
                    unsigned mp = mem_pairs;
                    while (mp) {
                        const int pr = __builtin_ctz(mp);
                        mp &= mp - 1;
                        const int term = pr >> 4, i = pr & 15;
                        const i32x4 dw = term ? dws[1] : dws[0];
                        const f32x4* hp = SPECTRAL ? bank_spec<BUCKETS>(p, dw.x, ch).hp + (size_t)i * blk_f4
                                                   : stash + (size_t)(term * p.stash_nbh + i) * blk_f4;
                        const f32x4* sp = p.spec + (size_t)(dw.y + (j - i - dw.z)) * blk_f4;
                        rows_product(hp + ti, sp + ti, ti, s, v);
                    }
                    item_store_inv(lds, s ? tw.i1 : tw.i0, ti + 1024 * s, v);
                }
                if (keep) in_mem |= 1u << last_new;
                // ---- 3. inverse passes
                if (!(kRowsAbl & 16)) simple_row_inv(lds, tw, ti, y);
                else {
                    lds_barrier();
#pragma unroll
                    for (int a = 0; a < 8; ++a) y[a] = lds[ti + 1024 * a];
                }
            } else {
#pragma unroll
                for (int a = 0; a < 8; ++a) y[a] = mk2(0.f, 0.f);
            }
            if (xf && j == 0) {
                if (round == 0) {                         // head of the row through the previous RIR: own slots, no barrier
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        if (tl + 1024 * a < kPrevPairs) s_prev[tl + 1024 * a] = y[a];
                    lds_barrier();                        // pass-1' reads of the buffer are over before round 1 writes it
                } else {
                    // crossfade(): x1[:, :n+1] * flip(arange(n+1)/n) + x2[:, :n+1] * (arange(n+1)/n), n = fade_len
                    const float fl = (float)p.fade_len;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int m = tl + 1024 * a, n0 = 2 * m;
                        if (n0 <= p.fade_len) {
                            const c32 yp = s_prev[m];
                            y[a].x = yp.x * ((float)(p.fade_len - n0) / fl) + y[a].x * ((float)n0 / fl);
                            if (n0 + 1 <= p.fade_len)
                                y[a].y = yp.y * ((float)(p.fade_len - n0 - 1) / fl) + y[a].y * ((float)(n0 + 1) / fl);
                        }
                    }
                }
            }
            }                                             // round
            int row_j = row;                              // (output addresses are rebuilt per block, in scalar registers:
            SSK_OPAQUE_S(row_j);                          //  hoisted out of the j loop they lived in VGPRs and spilled)
            if ((j < p.nb_y || j == 0) && part == 0) store_row_block(p, tl, (size_t)row_j, j, y);
            const bool last = j == nb_rows - 1;
            const int b1 = last ? p.t4 : min(p.t4, pooled_blocks_complete(kB * (j + 1)));
            // Short steps (SS2.0: 0.25 s of a 1-s row): pooled blocks that are exactly zero (live_blocks) are written, not
            // computed (at 44.1 kHz a 0.25-s step has 18 live pooled blocks of 69; the other 51 were three quarters of the STFT work).
            const int b_zero = live_blocks(p.n_valid, p.out_len, p.t4);
            const int b1c = min(b1, max(b0, b_zero));     // [b0, b1c) computed, [b1c, b1) zero
#if !defined(SS_ROWS_NO_L2_WARM)
            // what block j + 1 will load cold from HBM - term 0's RIR block j + 1 (1-s clips: exactly its new pair), or that
            // block's spectrum on the spectral bank - is pulled into L2 under this block's STFT phase
            // (time-domain bank: -1.3 ... -1.7 % in three alternating runs at 64 / 128 / 512 units; the same for the spectral
            // bank's block spectra measured +2.5 % at 128 / 512 units and is not done: profiles/r6/kbench_l2warm_44k.txt)
            if (!SPECTRAL && !XFADE && j + 1 < p.nb_y && dws[0].x >= 0 && j + 1 < nbh0) {
                const BankRow br = bank_row<BUCKETS>(p, dws[0].x, ch);
                if (br.es == 1) l2_touch(br.h + (size_t)(j + 1) * kB, min(kB, br.cap - (j + 1) * kB) * 4, tl, s_dummy);
            }
#endif
            if (kRowsAbl & 1) {
                if (y[0].x == 123.456f) p.sgram[0] = y[7].y;      // keeps the convolution alive
                lds_barrier();
            } else if (b1c > b0) {
                rows_stft_phase(lds, p, tl, row_j >> 1, row_j & 1, j, b0, b1c, last && b1c == b1, y, s_win, s_tw512, s_wq, s_res, s_tail,
                                s_tail, part);
            } else {
                lds_barrier();                            // (the phase's entry barrier: pass-1' reads of the buffer are over)
            }
            if (b1c < b1 && part == 0) {
                float* o = p.sgram + ((size_t)(row_j >> 1) * kBins4 * p.t4) * 2 + (row_j & 1);
                const int nz = b1 - b1c;
                for (int e = tl; e < kBins4 * nz; e += kT) {
                    const int b = e / nz, k = b1c + (e - b * nz);
                    o[((size_t)b * p.t4 + k) * 2] = 0.f;
                }
            }
            b0 = b1;
        }
        lds_barrier();                                    // s_res / the scratches are reused by the next row
    }
}

// ---------------------------------------------------------------------------------------------
// k_obs_blocks (round 6): the fused observation of rows longer than one block for SMALL steps - the reference's own arrangement
// at its Replica rate: 5 envs per GPU at 44.1 kHz (ss_baselines/av_nav/config/audionav/replica/train_telephone/
// audiogoal_depth_ddppo.yaml:3, configs/audionav/av_nav/replica/audiogoal.yaml:18).  k_obs_rows gives a row ONE workgroup that
// renders its output blocks one after the other (44.1 kHz: three forward and three inverse transforms, six products, three STFT
// phases, a stash round trip: 51-72 us however few rows there are, on a chip that is 90 % idle).  Here every OUTPUT BLOCK of a
// row has a workgroup (or 2^parts_log2 of them) of its own, on a CU of its own:
//   workgroup (row, j):  Y_j = sum_i H'_i * S'_{j-i} exactly as k_conv<loop> / k_conv_spec<loop> render block j (time-domain
//     bank: the j + 1 forward transforms accumulate in registers - redundant across the row's workgroups, on CUs that would
//     idle; no stash) -> inverse -> the block's samples in registers -> the STFT phase of k_obs_rows for the pooled blocks
//     [b0_j, b1_j) that block j completes (rows_stft_phase; split further over the block's parts).
//   The STFT frames that straddle into block j start in block j - 1: its workgroup leaves its last <= 640 samples in global
//     memory (`tails`) and releases a flag; workgroup (row, j) acquires it just before its STFT phase - long after it was set,
//     block j - 1 has one transform / product less to do.  Flags carry the launch's epoch: nothing is reset between launches.
// The launcher only uses this kernel when the whole grid fits the chip at one workgroup per CU, so every workgroup a flag is
// waited for is running or about to; the wait is bounded anyway (flag_acquire).  Same arithmetic per output sample and per
// pooled column as k_obs_rows: identical results.  Preconditions (launcher): n_valid == out_len (SoundSpaces 1.0 rows), no
// cross-fade, 2 or 3 output blocks.
template <bool SPECTRAL>
__global__ __launch_bounds__(1024) void k_obs_blocks(ConvParams p, int n_rows, float* tails, int* flags, int epoch) {
    __shared__ c32 lds[16 * kWaveScratch > kLdsComplex ? 16 * kWaveScratch : kLdsComplex];
    __shared__ float s_win[kNfft];
    __shared__ c32 s_tw512[kTw512Lds];
    __shared__ float s_res[kRowsResFloats];
    __shared__ c32 s_wq[16];
    const int t = threadIdx.x;
    const ThreadTw tw = load_thread_tw(p.tb.twM, p.tb.twItem, t);
    {   // the STFT's tables (unconditional clamped loads: see k_conv); consumed behind the phase's first barriers
        const float win_v = p.tb.win[t & (kNfft - 1)];
        const c32 tw512_v = p.tb.tw512[t & 255];
        const c32 wq0 = p.tb.twM[64 * (t & 15)];
        if (t < kNfft) s_win[t] = win_v;
        if (t < 256) s_tw512[posN(t)] = tw512_v;
        if (t < 16) s_wq[t] = wq0;
    }
    const int nb_rows = (p.out_len + kB - 1) / kB;
    // slot = ((row, part), j), j fastest: the workgroups of a row sit next to each other (one XCD: the block spectra H'_i that
    // block j re-reads after block j - 1 come out of that L2, as in k_conv_spec)
    const int slot = row_slot(blockIdx.x, (int)gridDim.x, p.xcd_map);
    const int rp = __builtin_amdgcn_readfirstlane(slot / nb_rows), j = slot - rp * nb_rows;
    const int part = rp & ((1 << p.parts_log2) - 1), row = rp >> p.parts_log2;
    if (row >= n_rows) return;
    const int unit = row >> 1, ch = row & 1;
    i32x4 dws[2];
    uniform_load8(p.desc + 8 * unit, dws[0], dws[1]);
    if (p.n_terms < 2) dws[1].x = -1;
    const bool last = j == nb_rows - 1;
    const int b0 = j == 0 ? 0 : min(p.t4, pooled_blocks_complete(kB * j));
    const int b1 = last ? p.t4 : min(p.t4, pooled_blocks_complete(kB * (j + 1)));
    if (dws[0].x < 0 && dws[1].x < 0) {                   // silent unit (simulator.py:610-612): exact zeros; nobody waits for it
        if (part) return;                                 // (the row's other workgroups are silent too)
        if (p.out) {
            const int lo = kB * j, hi = min(p.out_len, kB * (j + 1));
            for (int n = lo + t; n < hi; n += kT) p.out[(size_t)row * p.out_len + n] = 0.f;
        }
        float* o = p.sgram + (size_t)unit * kBins4 * p.t4 * 2 + ch;
        const int nz = b1 - b0;
        for (int e = t; e < kBins4 * nz; e += kT) {
            const int b = e / nz, k = b0 + (e - b * nz);
            o[((size_t)b * p.t4 + k) * 2] = 0.f;
        }
        return;
    }
    c32 acc[2][8];
    c32 y[8];
    bool any = false;
#pragma unroll
    for (int term = 0; term < 2; ++term) {
        const i32x4 dw = term ? dws[1] : dws[0];
        const int ridx = dw.x;
        if (ridx < 0) continue;
        const int spec0 = dw.y, m_min = dw.z, m_cnt = dw.w;
        const int L = uniform_load(p.rir_len + ridx);
        int nbh = (L + kB - 1) / kB;
        if (SPECTRAL) {
            const BankSpec bs = bank_spec(p, ridx, ch);
            nbh = min(nbh, bs.h_blocks);
            for (int i = 0; i < nbh; ++i) {
                const int m = j - i;
                if (m < m_min || m >= m_min + m_cnt) continue;
                int tl = t;
                SSK_OPAQUE1(tl);
                spec_block_product(p.spec, tl, bs.hp + (size_t)i * (kSpecComplex / 2) + tl, spec0 + (m - m_min), any, acc);
                any = true;
            }
        } else {
            const BankRow br = bank_row(p, ridx, ch);
            nbh = min(nbh, (br.cap + kB - 1) / kB);
            for (int i = 0; i < nbh; ++i) {
                const int m = j - i;
                if (m < m_min || m >= m_min + m_cnt) continue;
                int tl = t;
                SSK_OPAQUE1(tl);                          // (per transform: see k_conv)
                if (!any) {
                    conv_block<false, false>(lds, p, tw, tl, br, L, i, spec0 + (m - m_min), acc);
                    any = true;
                } else {
                    lds_barrier();                        // the previous transform's item reads are done
                    conv_block<true, false>(lds, p, tw, tl, br, L, i, spec0 + (m - m_min), acc);
                }
            }
        }
    }
    if (any) {
        if (!SPECTRAL) lds_barrier();
        items_to_time(lds, tw, t, acc, y);
    } else {
#pragma unroll
        for (int a = 0; a < 8; ++a) y[a] = mk2(0.f, 0.f);
    }
    if (part == 0) store_row_block(p, t, (size_t)row, j, y);
    const size_t hand = (size_t)row * (nb_rows - 1);      // the row's hand-off slots: [row][j] = block j -> block j + 1
    BlockSync sync{(!last && part == 0) ? flags + hand + j : nullptr, j > 0 ? flags + hand + (j - 1) : nullptr, epoch};
    rows_stft_phase<true>(lds, p, t, unit, ch, j, b0, b1, last, y, s_win, s_tw512, s_wq, s_res,
                          j > 0 ? tails + (hand + (j - 1)) * kTailFloats : nullptr,
                          (!last && part == 0) ? tails + (hand + j) * kTailFloats : nullptr, part, sync);
}

// ---------------------------------------------------------------------------------------------
// k_intensity: av_wan Intensity sensor (ss_baselines/av_wan/avwan_sensors.py:91-100) on the audiogoal:
//   thr = 0.1 * max(x);  onset = min over ears of the first index with x > thr (0 if none);
//   out = mean( x[:, onset : onset+num_frame] ** 2 )      (mean over the samples that exist)
// One 256-thread workgroup per unit; block reductions (max, min index per ear, sum) = wave butterflies + 4 partials.
struct IntensityParams {
    const float* x;    // [N][2][len]
    float* out;        // [N]
    int len, num_frame;
};

// value of `v` held by lane (lane ^ mask) of the same wave
__device__ __forceinline__ float lane_xor(float v, int lane, int mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __shfl_xor(v, mask, 64);
#elif defined(SSK_HOSTSIM)
    return hostsim_lane_read(v, lane ^ mask);
#else
    return v;
#endif
}
// wave-level butterfly reductions (6 exchange steps, every lane ends with the result); OP in {0 max, 1 min, 2 sum}
template <int OP>
__device__ __forceinline__ float wave_reduce(float v, int lane) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float o = lane_xor(v, lane, m);
        v = OP == 0 ? fmaxf(v, o) : OP == 1 ? fminf(v, o) : v + o;
    }
    return v;
}
// block reduction for 256 threads = 4 waves: one wave-level reduction + 4 partials through LDS (two barriers instead of
// the eight of a shared-memory tree)
template <int OP>
__device__ __forceinline__ float block_reduce256(float v, int t, float* part) {
    v = wave_reduce<OP>(v, t & 63);
    __syncthreads();                       // `part` may still be read from the previous reduction
    if ((t & 63) == 0) part[t >> 6] = v;
    __syncthreads();
    const float a = part[0], b = part[1], c = part[2], d = part[3];
    return OP == 0 ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : OP == 1 ? fminf(fminf(a, b), fminf(c, d)) : (a + b) + (c + d);
}

__global__ __launch_bounds__(256) void k_intensity(IntensityParams p) {
    __shared__ float part[4];
    const int t = threadIdx.x;
    const float* x = p.x + (size_t)blockIdx.x * 2 * p.len;
    float m = -3.402823466e38f;
    for (int i = t; i < 2 * p.len; i += 256) m = fmaxf(m, x[i]);
    const float thr = 0.1f * block_reduce256<0>(m, t, part);
    int first[2];                                        // first index above the threshold, per ear (len = none)
    for (int c = 0; c < 2; ++c) {
        int f = p.len;
        for (int i = t; i < p.len; i += 256)
            if (x[c * p.len + i] > thr) { f = i; break; }
        // indices < 2^24 are exact in float: the min reduction runs on the same float butterfly
        const int g = (int)block_reduce256<1>((float)f, t, part);
        first[c] = g == p.len ? 0 : g;                   // np.argmax of an all-False row is 0
    }
    const int onset = min(first[0], first[1]);
    const int n = min(p.num_frame, p.len - onset);
    float acc = 0.f;
    for (int i = t; i < 2 * n; i += 256) { const float v = x[(i / n) * p.len + onset + (i % n)]; acc += v * v; }
    const float tot = block_reduce256<2>(acc, t, part);
    if (t == 0) p.out[blockIdx.x] = n > 0 ? tot / (2.f * n) : 0.f;
}

// ---------------------------------------------------------------------------------------------
// k_scatter_rows: the RIR miss path's last hop.  n staged rows in wav layout ([frames][2], the layout the files and the ray
// tracer's output have) -> the planar bank rows bank[slot][c][j], zero behind each row's own length up to `cap`; and the
// rows' lengths into the bank's length table.  `staged`, `slots`, `lens` may be PINNED HOST memory (device-mapped under
// ROCm): the kernel then pulls the samples over the host link itself - one launch instead of three copies, a transpose
// and two index_copy dispatches (61 us of host time per miss step, profiles/r5/NOTES.md section 2).  Frames beyond a row's
// length are not read at all.  Grid: (chunks of 512 frames, n); 256 threads x 2 frames: 16-byte loads, 8-byte stores.
struct ScatterRowsParams {
    const float* staged;       // [n][staged_stride] floats, row i = frames j < lens[i] as (L, R) pairs
    const int* slots;          // [n] bank rows
    const int* lens;           // [n] frames of row i (<= cap)
    float* bank;               // bank[slot * unit_stride + c * chan_stride + j]
    int* bank_len;             // [slots of the bank] (may be nullptr)
    long long staged_stride, unit_stride;
    int chan_stride, cap;
};

__global__ __launch_bounds__(256) void k_scatter_rows(ScatterRowsParams p) {
    const int i = blockIdx.y;
    const int slot = p.slots[i];
    const int len = min(max(p.lens[i], 0), p.cap);
    const int j = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.bank_len) p.bank_len[slot] = len;
    if (j >= p.cap) return;
    float* dl = p.bank + (size_t)slot * p.unit_stride;
    float* dr = dl + p.chan_stride;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* src = p.staged + (size_t)i * p.staged_stride + 2 * (size_t)j;
    const bool al16 = ((p.staged_stride & 3) | (reinterpret_cast<size_t>(p.staged) & 15)) == 0;
    if (j + 1 < len) {
        if (al16) v = *reinterpret_cast<const f32x4*>(src);
        else { v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3]; }
    } else if (j < len) { v.x = src[0]; v.y = src[1]; }
    if (j + 1 < p.cap && ((p.chan_stride | p.unit_stride | (reinterpret_cast<size_t>(p.bank) >> 2)) & 1) == 0) {
        *reinterpret_cast<c32*>(dl + j) = mk2(v.x, v.z);
        *reinterpret_cast<c32*>(dr + j) = mk2(v.y, v.w);
    } else {
        dl[j] = v.x; dr[j] = v.y;
        if (j + 1 < p.cap) { dl[j + 1] = v.z; dr[j + 1] = v.w; }
    }
}

}  // namespace ssk

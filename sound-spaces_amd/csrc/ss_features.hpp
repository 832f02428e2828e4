// ss_features.hpp — k_features: ONE pass over the binaural waveform for every STFT-derived feature of BASELINE.json
// configs[4] ("GCC-PHAT + log-mel fused sensor"; extensions: the reference has neither) and, optionally, the
// reference's own pooled spectrogram (SpectrogramSensor.compute_spectrogram, soundspaces/tasks/nav.py:86-100).
//
// Until round 4 the three features were three kernels (k_spectrogram, k_logmel, k_gccphat), each reading the [N,2,len]
// waveform from HBM and redoing framing -> window -> 512-point rFFT of every (unit, ear, frame).  Here a wave owns four
// frames of BOTH ears (the cross-spectrum of GCC-PHAT needs the two ears in one place), transforms each ear once and
// keeps the Hermitian-split spectra X[k] of both ears in registers (2 x 32 VGPRs); from those it derives
//   * the power spectrum -> band-sparse mel filter bank -> log                       (log-mel, out [N, n_mels, T, 2])
//   * |X| -> mean over 4 bins x 4 frames -> log1p                                    (spectrogram, out [N, 65, T4, 2])
//   * X_l conj(X_r) / (|.| + eps) -> Hermitian merge -> inverse 256-point FFT -> lags (GCC-PHAT, out [N, 2 max_lag + 1, T])
// Same workgroup shape and staging as k_gccphat (256 threads = 4 waves, 16 frames of one unit per round, both padded
// segments staged in LDS by coalesced loads, next segment prefetched under the math), ONE scratch region per wave (tile,
// natural-order spectrum, power spectrum and pooled sums take turns in it: every stage pulls what it needs into registers
// before the next one writes), results collected in LDS and written as contiguous rows.
//
// Mel filter bank: the ABI's band-sparse table (start[j], w[j][max_len]) is padded to the WIDEST band; triangular mel
// filters widen with frequency, so lanes of one pass (bands 16 g .. 16 g + 15) have similar widths: the pass runs
// glen[g] = the widest band of ITS group (computed once per workgroup from the table), not max_len - 14 instead of 60
// 4-bin steps per lane and frame on the standard 64-band bank.
#pragma once
#include "ss_kernels.hpp"

namespace ssk {

constexpr int kFeatMaxMels = 64, kFeatMaxLen = 64;          // more bands take the stand-alone k_logmel (start <= 256: start + 64 <= kPowStride)
constexpr int kFeatMelTable = 3072;                         // floats: n_mels * max_len (64 x 36, 40 x 52, 32 x 64 ... fit); 12 KiB
static_assert(kGccMaxLag <= 32, "k_features extracts lags from output slots 0, 1 and 15 only");
constexpr int kGccResStride = kSegFrames + 1;
// PHAT cross-spectrum V[k], k < 256, of one frame in LDS: written by lane (f, q) at k = 4 (q + 16 i) + e and its mirror
// 256 - k (ds_write_b64: 16 contiguous lanes per group over 32 banks -> consecutive q must be consecutive slots), read
// back at k = q + 16 j for the inverse FFT (ds_read_b64: 32 lanes = two frames over 64 banks).  The natural order with
// posN padding was 2.1-way conflicted on the stores and 2-way on the reads; posV is free of both (tests/test_lds_banks.py).
__host__ __device__ constexpr int posV(int k) { return (k & 3) * 68 + (k >> 2); }
constexpr int kVStride = 272;                               // complex per frame: >= 4 * 68, = 16 mod 32
static_assert(4 * kVStride <= kWaveScratch && posV(255) < kVStride, "the four V frames live in the wave's scratch");
constexpr int kFeatSegComplex = kSegLen / 2;                // one ear's staged segment as c32 (2912 floats = 11 648 B)

struct FeatParams {
    const float* x;        // [N][2][len]
    float* sgram;          // [N][65][t4][2] or nullptr
    float* mel;            // [N][n_mels][n_frames][2] or nullptr
    float* gcc;            // [N][2*max_lag+1][n_frames] or nullptr
    Tables tb;
    const int* mel_start;  // [n_mels]
    const float* mel_w;    // [n_mels][max_len]
    int len, n_frames, t4, pad_mode, n_units;
    int n_mels, max_len, max_lag;
    float mel_eps, gcc_eps;
};

// Hermitian-split spectrum of this lane's 16 bins of one ear: X[4 i + e] = 2 X[k], Y[4 i + e] = conj(2 X[256 - k]),
// k = 4 b + e, b = q + 16 i (the pairing of stft_block); z128 = Z[128] (q == 0 only; X[128] = conj(Z[128]))
struct EarXY { c32 X[8], Y[8]; c32 z128; };

// 256-point FFT of the packed frame held as x[j] = frame[q + 16 j] -> EarXY.  `sc` = the wave's scratch (free on return).
__device__ __forceinline__ void ear_spectrum(c32* sc, int lane, c32 wq, const c32* tw512, c32 (&x)[16], EarXY& E) {
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
    for (int s = 0; s < 16; ++s) lds_st(fn + q + posN(16 * s), x[s]);
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
    E.z128 = mk2(0.f, 0.f);
    if (q == 0) E.z128 = fn[posN(128)];
    wave_sync();                                          // every lane holds its bins: the scratch is free again
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const c32 ptop = i == 0 ? (q == 0 ? k01[0].xy : prev0) : (q == 0 ? prev0 : prev1);
        const c32 zk[4] = {k01[i].xy, k01[i].zw, k23[i].xy, k23[i].zw};
        const c32 zp[4] = {ptop, p23[i].zw, p23[i].xy, p01[i].zw};
        const c32 ww[4] = {w01[i].xy, w01[i].zw, w23[i].xy, w23[i].zw};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const c32 P = add_conj(zk[e], zp[e]), Q = sub_conj(zk[e], zp[e]);
            const c32 wQ = cmul(Q, ww[e]);
            E.X[4 * i + e] = add_mi(P, wQ);
            E.Y[4 * i + e] = add_pi(P, wQ);
        }
    }
}

// power (mel) and pooled magnitude (spectrogram) of one ear from its split spectrum
template <class MELSTORE, class SGSTORE>
__device__ __forceinline__ void ear_power_features(c32* sc, int lane, const EarXY& E, bool want_mel, bool want_sg, int n_mels,
                                                   int max_len, float mel_eps, const float* s_w, const int* s_start,
                                                   const int* s_glen, MELSTORE mel_store, SGSTORE sg_store) {
    const int f = lane >> 4, q = lane & 15;
    float* pw = reinterpret_cast<float*>(sc) + f * kPowStride;
    float dsum[2], m0[2], m123[2];
    float px[2][4], py[2][4];                             // |2 X[k]|^2, |2 X[256 - k]|^2, k = 4 b + e
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float d = 0.f, m = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const c32 X = E.X[4 * i + e], Y = E.Y[4 * i + e];
            px[i][e] = fmaf(X.x, X.x, X.y * X.y);
            py[i][e] = fmaf(Y.x, Y.x, Y.y * Y.y);
            if (want_sg) {
                d += fast_sqrt(px[i][e]);
                const float my = fast_sqrt(py[i][e]);
                if (e == 0) m0[i] = 0.5f * my; else m += my;
            }
        }
        dsum[i] = 0.5f * d;
        m123[i] = 0.5f * m;
    }
    const float p128 = E.z128.x * E.z128.x + E.z128.y * E.z128.y;
    if (want_mel) {
        // Power spectrum of the 4 frames in natural order, 16-byte stores at a 16-byte lane stride (bank-conflict free; as 32
        // scalar stores per lane the four frames - kPowStride = 0 mod 32 - and lanes q, q + 8 met in the same banks: 35 % of
        // the kernel's LDS cycles were conflict cycles).  Direct half: bins 4 b .. 4 b + 3.  Mirror half: bins 256 - 4 b - e sit
        // in the aligned quad [252 - 4 b, 256 - 4 b) except e = 0, which is the FIRST float of the quad of lane b - 1; so lane b
        // writes {e = 0 of lane b + 1 (DPP), its e = 3, 2, 1}; lane 15 of i = 0 takes lane 0's i = 1 value, lane 15 of i = 1 the
        // bin 128 of lane 0; bin 256 (lane 0, i = 0, e = 0) goes out on its own.
        const float n0 = row_rol1(py[0][0], lane), n1 = row_rol1(py[1][0], lane), n128 = row_rol1(q == 0 ? 4.f * p128 : 0.f, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = q + 16 * i;
            *reinterpret_cast<f32x4*>(pw + 4 * b) = f32x4{0.25f * px[i][0], 0.25f * px[i][1], 0.25f * px[i][2], 0.25f * px[i][3]};
            const float nxt = i == 0 ? (q == 15 ? n1 : n0) : (q == 15 ? n128 : n1);
            *reinterpret_cast<f32x4*>(pw + 252 - 4 * b) = f32x4{0.25f * nxt, 0.25f * py[i][3], 0.25f * py[i][2], 0.25f * py[i][1]};
        }
        if (q == 0) pw[256] = 0.25f * py[0][0];
        for (int k = 257 + q; k < kPowStride; k += 16) pw[k] = 0.f;
        wave_sync();
        for (int g = 0; 16 * g < n_mels; ++g) {
            const int j = q + 16 * g;
            if (j < n_mels) {
                // band starts and max_len are multiples of 4 (ABI contract): two aligned 16-byte LDS reads per 4 bins
                const f32x4* pj = reinterpret_cast<const f32x4*>(pw + s_start[j]);
                const f32x4* wj = reinterpret_cast<const f32x4*>(s_w + j * max_len);
                c32 acc = mk2(0.f, 0.f), acc2 = mk2(0.f, 0.f);
                const int steps = s_glen[g];
                int i = 0;
                for (; i + 1 < steps; i += 2) {                            // two 4-bin steps in flight: four loads, then the math
                    const f32x4 a = wj[i], b = pj[i], a2 = wj[i + 1], b2 = pj[i + 1];
                    acc = acc + mk2(a.x * b.x, a.y * b.y);                 // (v_pk_fma_f32: two bins per instruction)
                    acc2 = acc2 + mk2(a2.x * b2.x, a2.y * b2.y);
                    acc = acc + mk2(a.z * b.z, a.w * b.w);
                    acc2 = acc2 + mk2(a2.z * b2.z, a2.w * b2.w);
                }
                if (i < steps) {
                    const f32x4 a = wj[i], b = pj[i];
                    acc = acc + mk2(a.x * b.x, a.y * b.y);
                    acc2 = acc2 + mk2(a.z * b.z, a.w * b.w);
                }
                acc = acc + acc2;
#if defined(__HIP_DEVICE_COMPILE__)
                mel_store(j, f, __builtin_amdgcn_logf(acc.x + acc.y + mel_eps) * 0.69314718055994531f);
#else
                mel_store(j, f, logf(acc.x + acc.y + mel_eps));
#endif
            }
        }
        wave_sync();                                      // the power spectra are dead
    }
    if (want_sg) {                                        // as the tail of stft_block: partial sums -> gather -> log1p
        float* ps = reinterpret_cast<float*>(sc);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ps[f * kPsStride + q + 16 * i] = dsum[i];
            ps[f * kPsStride + 32 + q + 16 * i] = m0[i];
            ps[f * kPsStride + 64 + q + 16 * i] = m123[i];
        }
        if (q == 0) ps[f * kPsStride + 96] = fast_sqrt(p128);
        if (q == 1) ps[f * kPsStride + 97] = 0.f;
        wave_sync();
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
        sg_store(r, fast_log1p(v * (1.0f / 16.0f)));
        if (lane == 0) {
            float v64 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) v64 += ps[g * kPsStride + 32];
            sg_store(64, fast_log1p(v64 * (1.0f / 16.0f)));
        }
        wave_sync();
    }
}

template <bool MEL, bool SG, bool GCC>
__global__ __launch_bounds__(256, 2) void k_features(FeatParams p) {     // two workgroups per CU (LDS): <= 256 VGPRs
    // [right-ear segment | 4 wave scratches]; the LEFT-ear segment is staged over the first scratches and is dead once every
    // wave has pulled its left frames, the right-ear segment stays readable until the waves get to it: only ONE ear's
    // frames (32 VGPRs) wait in registers while the other ear is transformed.  79.7 KiB of LDS in all: two workgroups per CU
    alignas(16) __shared__ c32 sc[kFeatSegComplex + 4 * kWaveScratch];
    __shared__ float res_mel[kFeatMaxMels * 33];            // [n_mels][16 frames][2 ears], rows padded to 33 floats (banks)
    // [lag][16 frames], rows 17 floats apart: lane (f, q) stores lag 2 q + u of frame f - at a 16-float pitch all sixteen q of a
    // frame met in ONE bank (32 LDS cycles per store instead of 2: the largest single source of the kernel's conflict cycles)
    __shared__ float res_gcc[(2 * kGccMaxLag + 1) * kGccResStride];
    __shared__ float res_sg[kBins4 * 8];                    // [65][4 blocks][2 ears]
    __shared__ float s_win[kNfft];
    alignas(16) __shared__ c32 s_tw512[kTw512Lds];
    alignas(16) __shared__ float s_w[kFeatMelTable];
    __shared__ int s_start[kFeatMaxMels];
    __shared__ int s_glen[kFeatMaxMels / 16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, q = lane & 15;
    // Rounds: task = (unit, group of 16 frames), task id = unit * groups + group.  Workgroup b takes tasks b, b + G, b + 2G ...
    // (G = gridDim.x <= two per CU): 1792 tasks on 512 workgroups = 4 rounds for the first 256 workgroups and 3 for the
    // others - one of each kind per CU - instead of 4 for everybody when a workgroup owned a contiguous range of one unit's groups
    const int groups = (p.n_frames + kSegFrames - 1) / kSegFrames;
    const int n_tasks = p.n_units * groups, G = (int)gridDim.x;
    constexpr bool want_mel = MEL, want_sg = SG, want_gcc = GCC;
    const int n_lags = 2 * p.max_lag + 1;
    f32x4* seg4 = reinterpret_cast<f32x4*>(sc);
    const float* seg_r = reinterpret_cast<const float*>(sc);                 // right ear: quads [0, 728)
    const float* seg_l = seg_r + kSegLen;                                    // left ear: over the scratches
    // Staging of a round's two segments (16 frames span kSegLen = 2912 samples per ear, from sample 2560 g - 256): every
    // thread fetches 6 quads with UNCONDITIONAL loads from addresses clamped into the row (one 16-byte load each when rows
    // are 16-byte aligned and len % 4 == 0, else four 4-byte loads), so the fetch of round g + 1 travels under round g's
    // math in every instantiation and on every group.  librosa's centre padding (positions before sample 0 / from sample
    // len on: reflect or zeros) is patched in LDS afterwards, by the first / last group of a row only (<= 256 + 512 samples
    // per ear).  (The helper of the stand-alone kernels resolves the padding per lane at load time: a ~300-instruction
    // edge path with dependent scalar loads, inlined at every call site.)
    const int len = p.len;
    const bool vec_ok = !(len & 3) && !(reinterpret_cast<size_t>(p.x) & 15);        // (rows are 2 len floats apart)
    f32x4 r[6];
    auto fetch = [&](int task) {
        const int unit = task / groups, g = task - unit * groups;
        const float* row0 = p.x + (size_t)unit * 2 * len;
        const int s0 = kHop * kSegFrames * g - kNfft / 2;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int e4 = t + 256 * k;                     // quad of the two-ear segment: ear = e4 >= 728
            const int e = e4 < 2 * kSegQuads ? e4 : 0;
            const int c = e >= kSegQuads, n = s0 + 4 * (e - c * kSegQuads);
            const float* row = row0 + (size_t)c * len;
            if (vec_ok) {
                r[k] = *reinterpret_cast<const f32x4*>(row + min(max(n, 0), len - 4));
            } else {
                r[k] = f32x4{row[min(max(n, 0), len - 1)], row[min(max(n + 1, 0), len - 1)], row[min(max(n + 2, 0), len - 1)],
                             row[min(max(n + 3, 0), len - 1)]};
            }
        }
    };
    for (int e = t; e < kNfft; e += 256) s_win[e] = p.tb.win[e];
    s_tw512[posN(t)] = p.tb.tw512[t];
    if (t < kFeatMaxMels / 16) s_glen[t] = 0;
    if ((int)blockIdx.x < n_tasks) fetch(blockIdx.x);
    if (want_mel) {
        if (t < p.n_mels) s_start[t] = p.mel_start[t];
        for (int e = t; e < p.n_mels * p.max_len; e += 256) s_w[e] = p.mel_w[e];      // (independent loads: pipelined)
        lds_barrier();
        if (t < p.n_mels) {                                 // quads the band spans; the widest band of a group of 16 sets
            int used = p.max_len >> 2;                      // the group's step count (module comment)
            const f32x4* wj = reinterpret_cast<const f32x4*>(s_w + t * p.max_len);
            while (used > 0) {
                const f32x4 w = wj[used - 1];
                if (w.x != 0.f || w.y != 0.f || w.z != 0.f || w.w != 0.f) break;
                --used;
            }
            atomicMax(&s_glen[t >> 4], used);
        }
    }
    c32 wq = p.tb.twM[64 * q];
    SSK_OPAQUE2(wq);                                        // see k_spectrogram
    const float eps4 = 4.f * p.gcc_eps;                     // the split yields 2X, so the products carry a factor 4
    for (int task = blockIdx.x; task < n_tasks; task += G) {
        const int unit = task / groups, g = task - unit * groups;
        const float* row0 = p.x + (size_t)unit * 2 * len;
        c32 xl[16];
        const int fl = 4 * wv + (lane >> 4);                // frame of this lane within the group
        {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int e4 = t + 256 * k;                 // right ear (quads >= 728) first in LDS, left ear over the scratches
                if (e4 < 2 * kSegQuads) seg4[e4 < kSegQuads ? e4 + kSegQuads : e4 - kSegQuads] = r[k];
            }
            const int s0 = kHop * kSegFrames * g - kNfft / 2;
            if (s0 < 0 || s0 + kSegLen > len) {             // (workgroup-uniform) a group that touches the row's ends:
                lds_barrier();                              // patch the positions outside [0, len)
                float* segw = reinterpret_cast<float*>(sc);
                for (int i = t; i < 2 * kSegLen; i += 256) {
                    const int c = i >= kSegLen, n = s0 + i - c * kSegLen;
                    if (n >= 0 && n < len) continue;
                    int m = n < 0 ? -n : 2 * (len - 1) - n;                       // reflect, excluding the edge sample
                    const bool ok = p.pad_mode == 0 && m >= 0 && m < len;
                    const float v = ok ? row0[(size_t)c * len + m] : 0.f;        // (frames that reach further are not live)
                    segw[(c ? 0 : kSegLen) + (i - c * kSegLen)] = v;
                }
            }
            lds_barrier();
            const bool live = kSegFrames * g + fl < p.n_frames;
            const c32* w2 = reinterpret_cast<const c32*>(s_win) + q;
            const c32* yl = reinterpret_cast<const c32*>(seg_l + kHop * fl) + q;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const c32 w = lds_ld(w2 + 16 * j), a = lds_ld(yl + 16 * j);
                xl[j] = live ? mk2(w.x * a.x, w.y * a.y) : mk2(0.f, 0.f);
            }
            lds_barrier();                                  // left segment dead: the wave scratches overlay it
        }
        if (task + G < n_tasks) fetch(task + G);            // next round's segments in flight under this round's math
        if (kSegFrames * g + 4 * wv < p.n_frames) {         // wave-uniform: at least one live frame in this block
            c32* wsc = sc + kFeatSegComplex + wv * kWaveScratch;
            EarXY L, R;
            ear_spectrum(wsc, lane, wq, s_tw512, xl, L);
            if (want_mel || want_sg)
                ear_power_features(wsc, lane, L, want_mel, want_sg, p.n_mels, p.max_len, p.mel_eps, s_w, s_start, s_glen,
                                   [&](int j, int f, float v) { res_mel[j * 33 + (4 * wv + f) * 2] = v; },
                                   [&](int b, float v) { res_sg[b * 8 + wv * 2] = v; });
            {   // the right ear's frames, from its own (still intact) segment
                const bool live = kSegFrames * g + fl < p.n_frames;
                const c32* w2 = reinterpret_cast<const c32*>(s_win) + q;
                const c32* yr = reinterpret_cast<const c32*>(seg_r + kHop * fl) + q;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const c32 w = lds_ld(w2 + 16 * j), b = lds_ld(yr + 16 * j);
                    xl[j] = live ? mk2(w.x * b.x, w.y * b.y) : mk2(0.f, 0.f);
                }
            }
            ear_spectrum(wsc, lane, wq, s_tw512, xl, R);
            if (want_gcc) {
                // G[k] = X_l[k] conj(X_r[k]), PHAT-weighted; V = Hermitian merge; g = 256-point inverse FFT of V (k_gccphat)
                c32* vn = wsc + (lane >> 4) * kVStride;
                f32x4 w01[2], w23[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x4* w4 = reinterpret_cast<const f32x4*>(s_tw512 + posN(4 * (q + 16 * i)));
                    w01[i] = w4[0]; w23[i] = w4[1];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int b = q + 16 * i;
                    const c32 ww[4] = {w01[i].xy, w01[i].zw, w23[i].xy, w23[i].zw};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        c32 gk = phat(cmulc(L.X[4 * i + e], R.X[4 * i + e]), eps4);            // G[k]
                        const c32 gpc = phat(cmulc(L.Y[4 * i + e], R.Y[4 * i + e]), eps4);     // conj(G[256-k])
                        c32 gp = mk2(gpc.x, -gpc.y);
                        herm_inv(gk, gp, ww[e]);                                              // -> 2 V[k], 2 V[256-k]
                        vn[posV(4 * b + e)] = gk;
                        if (4 * b + e != 0) vn[posV(256 - 4 * b - e)] = gp;                   // V[256] does not exist
                    }
                }
                if (q == 0) {                               // k = 128 pairs with itself: X[128] = conj(Z[128])
                    const c32 g128 = phat(cmulc(mk2(L.z128.x, -L.z128.y), mk2(R.z128.x, -R.z128.y)), p.gcc_eps);
                    vn[posV(128)] = mk2(2.f * g128.x, -2.f * g128.y);                         // 2 V[128] = 2 conj(G[128])
                }
                wave_sync();
                c32 x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = lds_ld(vn + posV(q) + 4 * j);            // posV(q + 16 j)
                fft16<true>(x);
                c32 wqi = wq;
                SSK_OPAQUE2(wqi);
                twiddle16<true>(x, wqi);
                wave_sync();                                // every lane has read V: the tiles may overwrite it
                c32* fr = wsc + (lane >> 4) * kFrameStride;
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2) lds_st(fr + r2 * 17 + q, x[r2]);
                wave_sync();
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2) x[r2] = lds_ld(fr + q * 17 + r2);
                wave_sync();                                // the tiles are read: the scratch is free for the right ear's features
                fft16<true>(x);
                constexpr float inv = 1.0f / 512.0f;
                // lags |tau| <= max_lag <= 32 only: tau = 2 (q + 16 s2) + u is <= 33 for s2 in {0, 1} and >= 480 for s2 = 15
#pragma unroll
                for (int s2i = 0; s2i < 3; ++s2i) {
                    const int s2 = s2i == 2 ? 15 : s2i;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int tau = 2 * (q + 16 * s2) + u;                    // lag 0..511; >= 256 means tau - 512
                        const int i = tau <= p.max_lag ? tau + p.max_lag : tau >= kNfft - p.max_lag ? tau - kNfft + p.max_lag : -1;
                        if (i >= 0) res_gcc[i * kGccResStride + fl] = inv * (u ? x[s2].y : x[s2].x);
                    }
                }
            }
            // (the right ear's power features AFTER the cross-spectrum: the left ear's spectrum is dead by then)
            if (want_mel || want_sg)
                ear_power_features(wsc, lane, R, want_mel, want_sg, p.n_mels, p.max_len, p.mel_eps, s_w, s_start, s_glen,
                                   [&](int j, int f, float v) { res_mel[j * 33 + (4 * wv + f) * 2 + 1] = v; },
                                   [&](int b, float v) { res_sg[b * 8 + wv * 2 + 1] = v; });
        }
        lds_barrier();
        const int nf = min(kSegFrames, p.n_frames - kSegFrames * g);
        if (want_mel) {                                     // rows of 16 frames x 2 ears = 128 contiguous bytes
            float* o = p.mel + ((size_t)unit * p.n_mels * p.n_frames + kSegFrames * g) * 2;
            for (int e = t; e < p.n_mels * 32; e += 256) {
                const int j = e >> 5, c = e & 31;
                if (c < 2 * nf) o[(size_t)j * p.n_frames * 2 + c] = res_mel[j * 33 + c];
            }
        }
        if (want_gcc) {
            float* o = p.gcc + (size_t)unit * n_lags * p.n_frames + kSegFrames * g;
            for (int e = t; e < n_lags * kSegFrames; e += 256) {
                const int i = e >> 4, c = e & 15;
                if (c < nf) o[(size_t)i * p.n_frames + c] = res_gcc[i * kGccResStride + c];
            }
        }
        if (want_sg) {                                      // 65 rows x 4 (block, ear-pair) float2 = 32-byte runs
            for (int e = t; e < kBins4 * 4; e += 256) {
                const int b = e >> 2, c2 = e & 3;
                if (4 * g + c2 < p.t4)
                    *reinterpret_cast<c32*>(p.sgram + ((size_t)unit * kBins4 + b) * p.t4 * 2 + 8 * g + 2 * c2) =
                        *reinterpret_cast<const c32*>(res_sg + 8 * b + 2 * c2);
            }
        }
        lds_barrier();                                      // the result buffers and the scratch are reused by the next group
    }
}

}  // namespace ssk

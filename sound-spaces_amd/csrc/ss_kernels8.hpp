// ss_kernels8.hpp — HALF-ROW kernels of the 16 kHz audio-observation path (two 512-thread workgroups per CU).
//
//   k_source_windows8 : 16384-sample source window -> 8192-bin spectrum (kernel order of the 8192-point core)
//   k_conv_half       : one workgroup per (unit, ear, half j): convolution of the half + (fused) its 13 pooled STFT
//                       time blocks.  Reference: soundspaces/simulator.py:629-647 + soundspaces/tasks/nav.py:86-100.
//
// Partitioning.  out[t] = sum_k h[k] x[t0 + t - k], t < 16000, h of <= 16000 taps.  h is cut into partitions of kP = 8000
// taps, h_i = h[8000 i : 8000 (i+1)].  A workgroup produces the output range (E_j - 8385, E_j], E_0 = 8385, E_1 = 16000,
// as the alias-free tail (circular indices 7999 .. 16383) of
//     IFFT_16384( sum_i H_i * S_{j,i} ),   H_i = rFFT(h_i, zero padded),  S_{j,i} = rFFT( x[t0 + E_j - 8000 i - 16384 + n], n < 16384 )
// (overlap-save with hop = partition = 8000, so every partition of a block lands on the same circular indices).
// Why these two ranges: STFT frame f only weighs samples [160 f - 200, 160 f + 200) (hann(400) centred in the 512-point
// frame).  Half 0 = frames 0..51 needs samples < 8360 <= 8385; half 1 = frames 52..100 needs samples >= 8120 >= 7615.
// So 26 pooled time blocks split 13 / 13 and neither half needs a sample the other one computed: no hand-off at all.
// Cost: 3 forward + 2 inverse 8192-point FFTs per (unit, ear) instead of 1 + 1 16384-point ones (+16 % arithmetic), and
// up to 4 window spectra of 64 KiB per (sound, t0) instead of 1-2 of 128 KiB.  Gain: 68 KiB of LDS per workgroup, so two
// workgroups are resident per CU and one's memory / LDS phases overlap the other's arithmetic.
// Scope: sr = 16000, n_valid = out_len = 16000, RIR rows of <= 16000 taps (planar bank, even capacity), no distractor,
// no cross-fade: the SoundSpaces 1.0 headline shape.  Everything else stays on k_conv.
#pragma once
#include "ss_fft8k.hpp"
#include "ss_kernels.hpp"

namespace ssk8 {

using ssk::Tables;
using ssk::uniform_load;
using ssk::uniform_load4;
using ssk::i32x4;
using ssk::mk4;
using ssk::ld_stream;
using ssk::st_stream;
using ssk::kNfft;
using ssk::kHop;
using ssk::kBins4;
using ssk::kWaveScratch;
using ssk::kTw512Lds;
using ssk::posN;

constexpr int kE0 = kValid;                  // 8385: half 0 covers output samples [0, 8385)
constexpr int kHalfFrames = 52;              // frames per half (13 pooled blocks of 4)
constexpr int kHalfBlocks = 13;

// ---------------------------------------------------------------------------------------------------------------------
// k_source_windows8: one workgroup per window.  desc[w] = {src_offset, src_len, start, wrap(, out slot)}: window sample n
// (0 <= n < 16384) = x[start + n], zero outside [0, src_len) (wrap: indices >= src_len continue at the clip start).
// Output: 4096 f32x4 = S'/(8*8192) in kernel order: thread t, item s, half h (values 2h, 2h+1) at (2 s + h)*512 + t.
struct SrcParams8 {
    const float* src;
    const int* desc;
    f32x4* spec;
    Tables tb;
    int desc_stride;      // 4 or 5 (word 4 = output slot)
    float scale;
};

__global__ __launch_bounds__(512) void k_source_windows8(SrcParams8 p) {
    __shared__ c32 lds[kLds8];
    const int t = threadIdx.x, w = blockIdx.x;
    const int* d = p.desc + p.desc_stride * w;
    const float* x = p.src + __builtin_amdgcn_readfirstlane(d[0]);
    const int len = __builtin_amdgcn_readfirstlane(d[1]), start = __builtin_amdgcn_readfirstlane(d[2]);
    const int wrap = __builtin_amdgcn_readfirstlane(d[3]);
    const int slot = p.desc_stride > 4 ? __builtin_amdgcn_readfirstlane(d[4]) : w;
    const ThreadTw8 tw = load_thread_tw8(p.tb.twM, p.tb.twItem8, t);
    pass1_fwd8<false>(lds, tw.p1, t, [&](int m) {
        return mk2(ssk::src_sample(x, len, start + 2 * m, wrap), ssk::src_sample(x, len, start + 2 * m + 1, wrap));
    });
    fwd_passes8(lds, tw, t);
    f32x4* o = p.spec + (size_t)slot * (kSpec8 / 2) + t;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        c32 v[4];
        item_load_fwd8(lds, tw.it[s], t + 512 * s, v);
        o[(2 * s) * 512] = mk4(v[0] * p.scale, v[1] * p.scale);
        o[(2 * s + 1) * 512] = mk4(v[2] * p.scale, v[3] * p.scale);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct HalfParams {
    const f32x4* spec;          // [slots][4096] f32x4: window spectra (8192-point kernel order)
    const float* rir;           // planar bank [R][2][cap], rows zero beyond their length
    const int* rir_len;         // [R]
    const int* desc;            // [N][8]: {rir | -1, slot(j=0,i=0), slot(0,1), slot(1,0), slot(1,1), 0, 0, 0}; slot -1 = zero window
    float* out;                 // audiogoal [N][2][16000] or nullptr
    float* sgram;               // spectrogram [N][65][26][2] or nullptr
    Tables tb;
    long long rir_unit_stride;
    int rir_chan_stride, rir_cap;
    int out_len;                // 16000
    int n_frames, t4, pad_mode;
    int n_rows;                 // 2 * N
    int xcd_map;
    int dbg;                    // timing experiments only: early exit point (0 = full kernel)
    const f32x4* hspec;         // SPEC kernels: partition spectra [R][2][2][4096] f32x4 (k_source_windows8 over the bank, scale 1)
};

// the 1-s row's samples this workgroup owns, from the inverse FFT's registers: circular index n = 2 (t + 512 a) (+1)
// <-> output sample tau = n + off (off = E_j - 16384), alias-free for n >= 7999
template <bool FUSE, bool SPEC = false>
__global__ __launch_bounds__(512, 4) void k_conv_half(HalfParams p) {
    __shared__ c32 lds[FUSE && 8 * kWaveScratch > kLds8 ? 8 * kWaveScratch : kLds8];
    __shared__ float s_win[FUSE ? kNfft : 1];
    __shared__ c32 s_tw512[FUSE ? kTw512Lds : 1];
    const int t = threadIdx.x;
    // the first n_rows workgroups are the j = 0 halves, the next n_rows the j = 1 halves: a CU's two resident workgroups
    // are one of each kind (1 + 1 vs 2 + 1 FFTs), and both ears of a unit share an XCD (ssk::row_slot)
    const int j = blockIdx.x >= p.n_rows;
    const int slot = ssk::row_slot(blockIdx.x - j * p.n_rows, p.n_rows, p.xcd_map);
    const int unit = slot >> 1, ch = slot & 1;
    const int* d = p.desc + 8 * unit;
    // Register budget: two workgroups per CU = 16 waves = 128 VGPRs per thread, and the loop below carries the
    // accumulator (32) across the passes (x[16] + butterfly temporaries + twiddle chain ~ 80).  Hence: only the two
    // pass twiddles are kept for the whole kernel; the four item twiddles are re-fetched (L2 hits) at each item stage,
    // and the STFT constants are fetched after the convolution.  The other resident workgroup hides these latencies.
    ThreadTw8 tw;
    tw.p1 = p.tb.twM[2 * t];
    tw.p2 = p.tb.twM[32 * (t & 31)];
    const int ridx = uniform_load(d);
    if (p.dbg == 1) { if (ridx == -12345) p.sgram[0] = tw.p1.x + tw.p2.y; return; }
    c32 acc[4][4];
    bool any = false;
    if (ridx >= 0) {
        const int L = uniform_load(p.rir_len + ridx);
        const float* h = p.rir + (size_t)ridx * p.rir_unit_stride + (size_t)ch * p.rir_chan_stride;
        const int nparts = (L + kP - 1) / kP;
        for (int i = 0; i < nparts && i < 2; ++i) {
            const int wslot = uniform_load(d + 1 + 2 * j + i);
            if (wslot < 0) continue;
            int tl = t;
            SSK_OPAQUE1(tl);                            // see k_conv: keeps LICM from hoisting the body's addresses
            if (SPEC) {
                // spectral bank: H'_i straight from HBM in item order, times the window spectrum, no forward FFT, no LDS
                const f32x4* hp = p.hspec + (((size_t)ridx * 2 + ch) * 2 + i) * (kSpec8 / 2) + tl;
                const f32x4* sp = p.spec + (size_t)wslot * (kSpec8 / 2) + tl;
                f32x4 hv[8], sv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { hv[k] = ld_stream(hp + k * 512); sv[k] = sp[k * 512]; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const c32 hh[4] = {hv[2 * s].xy, hv[2 * s].zw, hv[2 * s + 1].xy, hv[2 * s + 1].zw};
                    const c32 w[4] = {sv[2 * s].xy, sv[2 * s].zw, sv[2 * s + 1].xy, sv[2 * s + 1].zw};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        c32 pr = cmul(hh[e], w[e]);
                        if (s == 0 && e == 0 && tl == 0) pr = mk2(hh[0].x * w[0].x, hh[0].y * w[0].y);
                        if (any) acc[s][e] += pr; else acc[s][e] = pr;
                    }
                }
                any = true;
                continue;
            }
            if (any) lds_barrier();                     // previous partition's item reads of layout B are done
            const c32* h2 = reinterpret_cast<const c32*>(h + i * kP);
            const int m_end = min(kP, p.rir_cap - i * kP) >> 1;
            pass1_fwd8<true>(lds, tw.p1, tl, [&](int m) { return m < m_end ? ld_stream(h2 + m) : mk2(0.f, 0.f); });
            const f32x4* sp = p.spec + (size_t)wslot * (kSpec8 / 2) + tl;
            lds_barrier();
            pass2_8<false>(lds, tw.p2, tl);
            lds_barrier();
            // window-spectrum values (L2 / MALL hits): items 0-1 in flight under pass 3, items 2-3 under items 0-1 (all
            // eight at once, on top of the accumulator of a second partition, exceed the 128-VGPR budget of 16 waves / CU)
            f32x4 sv[4], sw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = sp[k * 512];
            pass3_fwd8(lds, tl);
            lds_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) sw[k] = sp[(4 + k) * 512];
            c32 wit[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) wit[s] = p.tb.twItem8[tl + 512 * s];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c32 v[4];
                item_load_fwd8(lds, wit[s], tl + 512 * s, v);
                const f32x4 s0 = s < 2 ? sv[2 * s] : sw[2 * s - 4], s1 = s < 2 ? sv[2 * s + 1] : sw[2 * s - 3];
                const c32 w[4] = {s0.xy, s0.zw, s1.xy, s1.zw};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    c32 pr = cmul(v[e], w[e]);
                    if (s == 0 && e == 0 && tl == 0) pr = mk2(v[0].x * w[0].x, v[0].y * w[0].y);   // (X[0], X[8192]) real
                    if (any) acc[s][e] += pr; else acc[s][e] = pr;
                }
            }
            any = true;
        }
    }
    if (p.dbg == 2) { if (any && acc[0][0].x == 123.456f) p.sgram[0] = acc[3][3].y; return; }
    c32 x[16];
    if (any) {
        c32 wit[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) wit[s] = p.tb.twItem8[t + 512 * s];
        if (!SPEC) lds_barrier();                       // item reads done before the merged spectrum overwrites layout B
#pragma unroll
        for (int s = 0; s < 4; ++s) item_store_inv8(lds, wit[s], t + 512 * s, acc[s]);
        inv_passes8(lds, tw, t, x);
    } else {
#pragma unroll
        for (int a = 0; a < 16; ++a) x[a] = mk2(0.f, 0.f);
    }
    if (p.dbg == 3) { if (x[0].x == 123.456f) p.sgram[0] = x[15].y; return; }
    const int off = (j ? p.out_len : kE0) - kSeg;       // output sample = circular index + off
    if (p.out) {                                        // half 0 stores [0, 8000), half 1 stores [8000, 16000)
        float* orow = p.out + ((size_t)unit * 2 + ch) * p.out_len;
        const int lo = j ? kP : 0, hi = j ? p.out_len : kP;
#pragma unroll
        for (int a = 7; a < 16; ++a) {
            const int tau = 2 * (t + 512 * a) + off;    // even circular index -> tau odd for half 0 (off = -7999), even for half 1
            if (tau >= lo && tau < hi) orow[tau] = x[a].x;
            if (tau + 1 >= lo && tau + 1 < hi) orow[tau + 1] = x[a].y;
        }
    }
    if (!FUSE) return;
    s_win[t] = p.tb.win[t];
    if (t < 256) s_tw512[posN(t)] = p.tb.tw512[t];
    const c32 wq = p.tb.twM[64 * (t & 15)];
    // ---- STFT of this half's 52 frames: seg[i] = padded row sample (base + i), base = -256 (half 0) / 8064 (half 1),
    // so that local frame fl = f - 52 j is seg[160 fl .. 160 fl + 512)
    float* seg = reinterpret_cast<float*>(lds);
    const int base = j ? kHop * kHalfFrames - kNfft / 2 : -(kNfft / 2);
    const int seg_len = kHop * (kHalfFrames - 1) + kNfft;            // 8672 floats
    lds_barrier();                                      // pass-1' reads of layout A8 are done
    for (int a = 7; a < 16; ++a) {
        const int n = 2 * (t + 512 * a);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tau = n + u + off, i = tau - base;
            if (n + u >= kP - 1 && tau >= 0 && tau < p.out_len && i >= 0 && i < seg_len) seg[i] = u ? x[a].y : x[a].x;
        }
    }
    // everything of the segment the inverse FFT does not provide: the centre padding (reflect without the edge sample,
    // or zeros) and, in half 0, the zero-weighted tail [8385, 8416) of frame 51 (finite values are all that is needed)
    for (int i = t; i < seg_len; i += kT8) {
        const int tau = base + i;
        if (tau < 0 || tau >= p.out_len || (!j && tau >= kE0)) seg[i] = 0.f;
    }
    lds_barrier();
    if (p.pad_mode == 0 && t < kNfft / 2) {             // reflect: y[-1-u] = y[1+u], y[len+u] = y[len-2-u]
        if (!j) seg[kNfft / 2 - 1 - t] = seg[kNfft / 2 + 1 + t];
        else seg[p.out_len + t - base] = seg[p.out_len - 2 - t - base];
    }
    lds_barrier();
    const int lane = t & 63, wv = t >> 6;
    const bool two = wv + 8 < kHalfBlocks;
    const int f0 = kHalfFrames * j;                      // first global frame of this half
    c32 x0[16], x1[16];
    // a frame beyond the row's last (global frame >= n_frames) loads zeros: stft_load_padded's `tf >= n_frames` test is
    // done in local frame numbers against the local count
    ssk::stft_load_padded(seg, 4 * wv + (lane >> 4), p.n_frames - f0, lane & 15, s_win, x0);
    ssk::stft_load_padded(seg, 4 * (wv + 8) + (lane >> 4), two ? p.n_frames - f0 : 0, lane & 15, s_win, x1);
    lds_barrier();                                      // the segment is dead: the per-wave scratch overlays it
    float* o = p.sgram + (size_t)unit * kBins4 * p.t4 * 2;
    const int tb0 = kHalfBlocks * j + wv;
    ssk::stft_block(lds + wv * kWaveScratch, lane, wq, s_tw512, x0, [&](int b, float v) { o[(b * p.t4 + tb0) * 2 + ch] = v; });
    if (two) {
        ssk::wave_sync();
        ssk::stft_block(lds + wv * kWaveScratch, lane, wq, s_tw512, x1,
                        [&](int b, float v) { o[(b * p.t4 + tb0 + 8) * 2 + ch] = v; });
    }
}

}  // namespace ssk8

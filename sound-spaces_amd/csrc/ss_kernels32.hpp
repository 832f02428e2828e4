// ss_kernels32.hpp — the loop-free observation kernels on the 512-thread / 32-values-per-thread FFT core
// (ss_fft_core32.hpp): one (unit, ear) row per workgroup, one RIR block, one output block - SoundSpaces 1.0 at 16 kHz
// with 1-s clips (soundspaces/simulator.py:629-632) and every other step the launcher classifies as SIMPLE.
//
//   k_source_windows32 : source clip window -> block spectrum S' in THIS core's register order
//   k_conv32<FUSE,TAB> : RIR row -> audiogoal [N,2,out_len] and / or, fused, the spectrogram [N,65,T4,2]
//                        (reference: fftconvolve x2 + slice, simulator.py:629-632; compute_spectrogram, nav.py:86-100)
//
// Same convolution model, descriptors, bank addressing and STFT building blocks as ss_kernels.hpp; only the transform
// differs (see ss_fft_core32.hpp for why).  Per row: pass 1 | B | pass 2 | B | item stage (forward radix-16, Hermitian
// split, product with the window spectrum, merge, inverse radix-16: in place) | B | pass 2' | B | pass 1' -> registers.
#pragma once
#include "ss_fft_core32.hpp"
#include "ss_kernels.hpp"

namespace ssk {

constexpr int kSpec32Quads = kSpecComplex / 2;      // f32x4 per stored spectrum (8192), [i * 512 + q], i < 16

__global__ __launch_bounds__(512) void k_source_windows32(SrcParams p) {
    alignas(16) __shared__ c32 lds[kLds32Complex + kTwP2];
    const int t = threadIdx.x, w = blockIdx.x;
    const int* d = p.desc + p.desc_stride * w;
    const float* x = p.src + __builtin_amdgcn_readfirstlane(d[0]);
    const int len = __builtin_amdgcn_readfirstlane(d[1]), start = __builtin_amdgcn_readfirstlane(d[2]);
    const int wrap = __builtin_amdgcn_readfirstlane(d[3]);
    const int slot = p.desc_stride > 4 ? __builtin_amdgcn_readfirstlane(d[4]) : w;
    const ThreadTw32 tw = load_thread_tw32(p.tb.twM, p.tb.twG, t);
    c32* s_tw2 = lds + kLds32Complex;
    s_tw2[t] = p.tb.twP2[t];
    pass1_fwd32<false>(lds, tw.p1, t, [&](int m) {
        return mk2(src_sample(x, len, start + 2 * m, wrap), src_sample(x, len, start + 2 * m + 1, wrap));
    });
    lds_barrier();
    pass2_32<false>(lds, s_tw2, t);
    lds_barrier();
    c32 v[32];
    item32_load_fwd(lds, tw.g, t, v);
    const float scale = p.scale;
    f32x4* o = p.spec + (size_t)slot * kSpec32Quads + t;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i * 512] = mk4(v[2 * i] * scale, v[2 * i + 1] * scale);
}

// Fused STFT phase on 8 waves: as fused_stft_phase (ss_kernels.hpp), the row goes from registers into LDS with
// librosa's centre padding materialised around it; a wave owns the pooled time blocks wv, wv + 8, wv + 16, wv + 24,
// pulls the frames of all of them into registers (<= 128 VGPRs: this core has 256), and after one barrier runs them
// back to back in its private scratch with wave-scope synchronisation only.
__device__ __forceinline__ void fused_stft_phase32(c32* lds, const ConvParams& p, int t, int unit, int ch, const c32 (&y)[16],
                                                   const float* s_win, const c32* s_tw512, c32 wq, float* s_res) {
    float* yl = reinterpret_cast<float*>(lds) + kNfft / 2;          // sample 0 of the row
    const int len = p.out_len;
    lds_barrier();   // all pass-1' reads are done
    c32* yl2 = reinterpret_cast<c32*>(yl) + t;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const int n = 2 * (t + 512 * a);
        yl2[512 * a] = mk2(n < p.n_valid ? y[a].x : 0.f, n + 1 < p.n_valid ? y[a].y : 0.f);   // zeros beyond n_valid
    }
    lds_barrier();
    if (t < kNfft / 2) {                                  // the two pads: reflect (excluding the edge sample) or zeros
        const bool refl = p.pad_mode == 0;
        const float l = refl ? yl[1 + t] : 0.f, r = refl ? yl[len - 2 - t] : 0.f;
        yl[-1 - t] = l;
        yl[len + t] = r;
    }
    lds_barrier();
    const int lane = t & 63, wv = t >> 6;
    const int live = live_blocks(p.n_valid, len, p.t4);
    const float* padded = reinterpret_cast<const float*>(lds);   // frame tf = floats [160 tf, 160 tf + 512)
    c32 x0[16], x1[16], x2[16], x3[16];
    const bool b0 = wv < live, b1 = wv + 8 < live, b2 = wv + 16 < live, b3 = wv + 24 < live;
    stft_load_padded(padded, 4 * wv + (lane >> 4), b0 ? p.n_frames : 0, lane & 15, s_win, x0);
    stft_load_padded(padded, 4 * (wv + 8) + (lane >> 4), b1 ? p.n_frames : 0, lane & 15, s_win, x1);
    stft_load_padded(padded, 4 * (wv + 16) + (lane >> 4), b2 ? p.n_frames : 0, lane & 15, s_win, x2);
    stft_load_padded(padded, 4 * (wv + 24) + (lane >> 4), b3 ? p.n_frames : 0, lane & 15, s_win, x3);
    lds_barrier();                                          // the row is dead: the wave scratches overlay it
    c32* sc = lds + wv * kWaveScratch;
    if (b0) stft_block(sc, lane, wq, s_tw512, x0, [&](int b, float v) { s_res[b * p.t4 + wv] = v; });
    if (b1) { wave_sync(); stft_block(sc, lane, wq, s_tw512, x1, [&](int b, float v) { s_res[b * p.t4 + wv + 8] = v; }); }
    if (b2) { wave_sync(); stft_block(sc, lane, wq, s_tw512, x2, [&](int b, float v) { s_res[b * p.t4 + wv + 16] = v; }); }
    if (b3) { wave_sync(); stft_block(sc, lane, wq, s_tw512, x3, [&](int b, float v) { s_res[b * p.t4 + wv + 24] = v; }); }
    lds_barrier();
    float* o = p.sgram + (size_t)unit * kBins4 * p.t4 * 2 + ch;
    const int k = t & 31;                                   // (t4 <= 26 on this path)
    if (k < p.t4)
        for (int b = t >> 5; b < kBins4; b += kT32 / 32) o[2 * (b * p.t4 + k)] = k < live ? s_res[b * p.t4 + k] : 0.f;
}

constexpr int kConv32LdsComplex = kLds32Complex + kTwP2;
static_assert(8 * kWaveScratch <= kConv32LdsComplex, "the 8 wave scratches of the STFT phase overlay the FFT buffer");

template <bool FUSE, bool TAB>
__global__ __launch_bounds__(512) void k_conv32(ConvParams p, UnitTab<TAB> ut = UnitTab<TAB>()) {
    alignas(16) __shared__ c32 lds[kConv32LdsComplex];
    __shared__ float s_win[FUSE ? kNfft : 1];
    alignas(16) __shared__ c32 s_tw512[FUSE ? kTw512Lds : 1];
    __shared__ float s_res[FUSE ? kResFloats : 1];
    const int t = threadIdx.x;
    const int slot = row_slot(blockIdx.x, gridDim.x, p.xcd_map);
    const int unit = slot >> 1, ch = slot & 1;
    const int* d = p.desc + 8 * unit;
    c32* s_tw2 = lds + kLds32Complex;
    const ThreadTw32 tw = load_thread_tw32(p.tb.twM, p.tb.twG, t);
    const c32 tw2_v = p.tb.twP2[t];
    // fused path: table values are only FETCHED here; they go to LDS after the convolution (see k_conv)
    c32 wq = mk2(1.f, 0.f), tw512_v = mk2(0.f, 0.f);
    float win_v = 0.f;
    if (FUSE) {
        win_v = p.tb.win[t];
        tw512_v = p.tb.tw512[t & 255];
        wq = p.tb.twM[64 * (t & 15)];
    }
    c32 y[16];
    int ridx;
    if constexpr (TAB) ridx = ut.tab[kTabWords * unit]; else ridx = __builtin_amdgcn_readfirstlane(d[0]);
    bool active = false;
    if (ridx >= 0) {
        const float* h = p.rir + (size_t)ridx * p.rir_unit_stride + (size_t)ch * p.rir_chan_stride;
        const int es = p.rir_elem_stride, cap = p.rir_cap;
        const bool planar = es == 1 && !(cap & 1) && !(reinterpret_cast<size_t>(h) & 7);   // 8-byte aligned rows
        // the row's address needs only ridx: its loads go out BEFORE the length / window words are waited for
        c32 hraw[16];
        if (planar) {
            const c32* h2 = reinterpret_cast<const c32*>(h);
            const int m_end = cap >> 1;
#pragma unroll
            for (int a = 0; a < 16; ++a) hraw[a] = ld_stream(h2 + min(t + 512 * a, m_end - 1));   // clamped, masked where consumed
        }
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
                pass1_fwd32<true>(lds, tw.p1, t, [&](int m) { return m < m_end ? hraw[(m - t) >> 9] : mk2(0.f, 0.f); });
            } else {
                pass1_fwd32<true>(lds, tw.p1, t, [&](int m) {
                    const int n = 2 * m;
                    return mk2(n < cap ? h[(size_t)n * es] : 0.f, n + 1 < cap ? h[(size_t)(n + 1) * es] : 0.f);
                });
            }
            s_tw2[t] = tw2_v;
            const f32x4* sp = p.spec + (size_t)slot0 * kSpec32Quads + t;
            lds_barrier();
            // the window spectrum (L2 / MALL hits) travels under pass 2
            f32x4 sv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = sp[i * 512];
            pass2_32<false>(lds, s_tw2, t);
            lds_barrier();
            {
                c32 v[32];
                item32_load_fwd(lds, tw.g, t, v);
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    const c32 w = (e & 1) ? sv[e >> 1].zw : sv[e >> 1].xy;
                    c32 pr = cmul(v[e], w);
                    if (e == 0 && t == 0) pr = mk2(v[0].x * w.x, v[0].y * w.y);   // (X[0], X[16384]) are real
                    v[e] = pr;
                }
                item32_store_inv(lds, tw.g, t, v);
            }
            lds_barrier();
            pass2_32<true>(lds, s_tw2, t);
            lds_barrier();
            pass1_inv32(lds, tw.p1, t, y);
            active = true;
        }
    }
    if (!active) {
#pragma unroll
        for (int a = 0; a < 16; ++a) y[a] = mk2(0.f, 0.f);
    }
    int ounit = unit;                                       // (unit table: units are dealt out sorted; see kTabWords)
    if constexpr (TAB) ounit = ut.tab[kTabWords * unit + 2];
    if (p.out) {                                            // one output block: samples [0, n_valid), zeros up to out_len
        const size_t row = (size_t)ounit * 2 + ch;
        float* orow = p.out + row * p.out_len;
        const int nv = p.n_valid;
        if (!(nv & 1) && !(reinterpret_cast<size_t>(orow) & 7)) {
            c32* o2 = reinterpret_cast<c32*>(orow) + t;
            const int m_end = nv >> 1;
#pragma unroll
            for (int a = 0; a < 16; ++a) if (t + 512 * a < m_end) st_stream(o2 + 512 * a, y[a]);
        } else {
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                const int n = 2 * (t + 512 * a);
                if (n < nv) orow[n] = y[a].x;
                if (n + 1 < nv) orow[n + 1] = y[a].y;
            }
        }
        for (int n = p.n_valid + t; n < p.out_len; n += kT32) orow[n] = 0.f;
    }
    if (FUSE) {
        s_win[t] = win_v;                                   // visible to the STFT phase after its first barrier
        if (t < 256) s_tw512[posN(t)] = tw512_v;
        fused_stft_phase32(lds, p, t, ounit, ch, y, s_win, s_tw512, wq, s_res);
    }
}

}  // namespace ssk

// hostsim.cpp — TEST INFRASTRUCTURE: runs the HIP kernels of sound-spaces_amd/csrc on the host, one workgroup
// at a time, each thread a ucontext fiber, so tests/test_hostsim.py can compare them with the oracle on CPU.
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "hip_shim.h"
#include "../../sound-spaces_amd/csrc/ss_kernels.hpp"
#include "../../sound-spaces_amd/csrc/ss_kernels32.hpp"
#include "../../sound-spaces_amd/csrc/ss_features.hpp"
#include "../../sound-spaces_amd/csrc/ss_tables.hpp"

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {
constexpr size_t kStack = 128 * 1024;
ucontext_t g_main;
std::vector<ucontext_t> g_ctx;
std::vector<char> g_stacks;
std::vector<char> g_done;
int g_cur = 0, g_nthreads = 0;
long g_progress = 0;
std::function<void()> g_body;

struct Barrier { int count = 0; long generation = 0; };
Barrier g_block_barrier;
std::vector<Barrier> g_wave_barrier;

void yield_fiber() {
    const int me = g_cur;
    swapcontext(&g_ctx[me], &g_main);
    threadIdx.x = me;
}

void barrier_wait(Barrier& b, int expected) {
    const long gen = b.generation;
    if (++b.count == expected) { b.count = 0; ++b.generation; ++g_progress; return; }
    while (b.generation == gen) yield_fiber();
}

void fiber_entry() {
    g_body();
    g_done[g_cur] = 1;
    ++g_progress;
    swapcontext(&g_ctx[g_cur], &g_main);
}

// run one workgroup: fibers are resumed round-robin; barriers are generation counters
int run_block(int nthreads, const std::function<void()>& body) {
    g_body = body;
    g_nthreads = nthreads;
    g_ctx.assign(nthreads, ucontext_t{});
    if (g_stacks.size() < kStack * nthreads) g_stacks.resize(kStack * nthreads);
    g_done.assign(nthreads, 0);
    g_block_barrier = Barrier{};
    g_wave_barrier.assign((nthreads + 63) / 64, Barrier{});
    for (int i = 0; i < nthreads; ++i) {
        getcontext(&g_ctx[i]);
        g_ctx[i].uc_stack.ss_sp = g_stacks.data() + kStack * i;
        g_ctx[i].uc_stack.ss_size = kStack;
        g_ctx[i].uc_link = &g_main;
        makecontext(&g_ctx[i], fiber_entry, 0);
    }
    blockDim.x = nthreads;
    for (;;) {
        int live = 0;
        const long before = g_progress;
        for (int i = 0; i < nthreads; ++i) {
            if (g_done[i]) continue;
            g_cur = i;
            threadIdx.x = i;
            swapcontext(&g_main, &g_ctx[i]);
            live += !g_done[i];
        }
        if (!live) break;
        if (g_progress == before) {
            std::fprintf(stderr, "hostsim: deadlock (divergent barriers) with %d live threads\n", live);
            return -2;
        }
    }
    return 0;
}

ssk::Tables host_tables() {
    static std::vector<float> tab = ssk_host::build_tables();
    ssk::Tables tb;
    tb.twM = reinterpret_cast<const ssk::c32*>(tab.data() + ssk_host::kTwMOff);
    tb.twItem = reinterpret_cast<const ssk::c32*>(tab.data() + ssk_host::kTwItemOff);
    tb.tw512 = reinterpret_cast<const ssk::c32*>(tab.data() + ssk_host::kTw512Off);
    tb.win = tab.data() + ssk_host::kWinOff;
    tb.twG = reinterpret_cast<const ssk::c32*>(tab.data() + ssk_host::kTwGOff);
    tb.twP2 = reinterpret_cast<const ssk::c32*>(tab.data() + ssk_host::kTwP2Off);
    return tb;
}
}  // namespace

// a second length bucket for the next hs_conv / hs_obs_rows call (cleared by it): entries >= first live in `rir` [n,2,cap]
static int g_spec_n_valid = -1;
static int g_parts_log2 = 0;          // the next fused hs_conv / hs_conv_spec / hs_obs_rows call: 2^k workgroups per row (cleared by it)
static const float* g_b2_rir = nullptr;
static int g_b2_first = 0, g_b2_cap = 0;
static void apply_bucket2(ssk::ConvParams& p) {
    p.n_buckets = 1;
    for (auto& b : p.bk) b = ssk::BankBucket{nullptr, nullptr, 0x7fffffff, 0, 0, 0};
    if (g_b2_rir) {
        p.n_buckets = 2;
        p.bk[0] = ssk::BankBucket{g_b2_rir, nullptr, g_b2_first, g_b2_cap, (g_b2_cap + ssk::kB - 1) / ssk::kB, 0};
    }
    g_b2_rir = nullptr;
}

void hostsim_syncthreads() { barrier_wait(g_block_barrier, g_nthreads); }
void hostsim_wave_sync() { barrier_wait(g_wave_barrier[g_cur / 64], 64); }
static float g_xch[16][64];
float hostsim_lane_read(float v, int src_lane) {
    const int w = g_cur / 64;
    g_xch[w][g_cur % 64] = v;
    hostsim_wave_sync();
    const float r = g_xch[w][src_lane];
    hostsim_wave_sync();
    return r;
}

extern "C" {

void hs_set_bucket2(const float* rir, int first, int cap) { g_b2_rir = rir; g_b2_first = first; g_b2_cap = cap; }
// the next hs_spectrogram call: rows are known to be zero from sample n_valid on (the library's own two-launch path)
void hs_set_spec_n_valid(int n_valid) { g_spec_n_valid = n_valid; }
void hs_set_parts_log2(int k) { g_parts_log2 = k; }
// the next hs_obs_rows call runs k_obs_blocks (one workgroup per OUTPUT BLOCK of a row, tails handed over through memory)
static int g_obs_blocks = 0;
void hs_set_obs_blocks(int on) { g_obs_blocks = on; }

int hs_source_windows(const float* src, const int* desc, float* spec, int n_windows) {
    ssk::SrcParams p;
    p.src = src; p.desc = desc; p.spec = reinterpret_cast<ssk::f32x4*>(spec); p.tb = host_tables();
    p.desc_stride = 4; p.scale = ssk::kWindowScale;
    gridDim = dim3{(unsigned)n_windows, 1, 1};
    for (int w = 0; w < n_windows; ++w) {
        blockIdx = dim3{(unsigned)w, 0, 0};
        int rc = run_block(ssk::kT, [&] { ssk::k_source_windows(p); });
        if (rc) return rc;
    }
    return 0;
}

int hs_conv(int fuse, int simple, const float* spec, const float* rir, const int* rir_len, const int* desc, float* out,
            float* sgram, int n_units, long long us, int cs, int es, int cap, int n_valid, int out_len, int pad_mode,
            int persist) {
    ssk::ConvParams p;
    p.spec = reinterpret_cast<const ssk::f32x4*>(spec); p.rir = rir; p.rir_len = rir_len; p.desc = desc;
    p.out = out; p.sgram = sgram; p.tb = host_tables();
    p.rir_unit_stride = us; p.rir_chan_stride = cs; p.rir_elem_stride = es; p.rir_cap = cap;
    p.n_valid = n_valid; p.out_len = out_len;
    p.n_frames = 1 + out_len / ssk::kHop;
    p.t4 = (p.n_frames + 3) / 4;
    p.pad_mode = pad_mode;
    p.hspec = nullptr; p.h_blocks = 0; p.xcd_map = 0; p.stash = nullptr; p.stash_nbh = 0; p.stash_terms = 0; p.n_terms = 2; p.parts_log2 = 0;
    apply_bucket2(p);
    p.fade_len = static_cast<int>(0.05 * out_len);
    const bool xfade = simple == 2;                     // simple: 0 = loop kernel, 1 = SIMPLE, 2 = loop kernel + XFADE
    if (xfade) simple = 0;
    const int nb_y = n_valid == 0 ? 1 : (n_valid + ssk::kB - 1) / ssk::kB;
    // WIDE: a row longer than one block of which only block 0 is rendered (the library's wide_one_block_ok)
    const bool wide = fuse && out_len > ssk::kB && n_valid <= ssk::kB && ssk::live_blocks(n_valid, out_len, p.t4) <= 26;
    if (wide && simple) return -3;
    if (fuse && !wide && (nb_y != 1 || out_len > ssk::kB || p.t4 > 26)) return -1;
    if (persist > 0) {                                  // k_conv_rows: `persist` workgroups walk the 2*n_units rows
        if (fuse || !simple || nb_y != 1 || es != 1 || (cap & 1) || cap > ssk::kB) return -2;
        gridDim = dim3{(unsigned)persist, 1, 1};
        for (int b = 0; b < persist && b < 2 * n_units; ++b) {
            blockIdx = dim3{(unsigned)b, 0, 0};
            int rc = run_block(ssk::kT, [&] {
                ssk::k_conv_rows(p, 2 * n_units);
            });
            if (rc) return rc;
        }
        return 0;
    }
    p.nb_y = nb_y;
    p.parts_log2 = fuse && !wide ? g_parts_log2 : 0;
    g_parts_log2 = 0;
    const int gx = (2 * n_units) << p.parts_log2;
    gridDim = dim3{(unsigned)gx, (unsigned)nb_y, 1};
    for (int b = 0; b < gx * nb_y; ++b) {
        {
            blockIdx = dim3{(unsigned)(b % gx), (unsigned)(b / gx), 0};
            int rc = run_block(ssk::kT, [&] {
                if (wide) { if (xfade) ssk::k_conv<true, false, true, false, true>(p); else ssk::k_conv<true, false, false, false, true>(p); }
                else if (xfade) { if (fuse) ssk::k_conv<true, false, true>(p); else ssk::k_conv<false, false, true>(p); }
                else if (fuse) { if (simple) ssk::k_conv<true, true>(p); else ssk::k_conv<true, false>(p); }
                else { if (simple) ssk::k_conv<false, true>(p); else ssk::k_conv<false, false>(p); }
            });
            if (rc) return rc;
        }
    }
    return 0;
}

// ---- 512-thread core (ss_kernels32.hpp): window spectra in ITS order + the loop-free row kernel
int hs_source_windows32(const float* src, const int* desc, float* spec, int n_windows) {
    ssk::SrcParams p;
    p.src = src; p.desc = desc; p.spec = reinterpret_cast<ssk::f32x4*>(spec); p.tb = host_tables();
    p.desc_stride = 4; p.scale = ssk::kWindowScale;
    gridDim = dim3{(unsigned)n_windows, 1, 1};
    for (int w = 0; w < n_windows; ++w) {
        blockIdx = dim3{(unsigned)w, 0, 0};
        int rc = run_block(ssk::kT32, [&] { ssk::k_source_windows32(p); });
        if (rc) return rc;
    }
    return 0;
}

int hs_conv32(int fuse, const float* spec, const float* rir, const int* rir_len, const int* desc, float* out,
              float* sgram, int n_units, long long us, int cs, int es, int cap, int n_valid, int out_len, int pad_mode,
              int use_tab) {
    ssk::ConvParams p;
    p.spec = reinterpret_cast<const ssk::f32x4*>(spec); p.rir = rir; p.rir_len = rir_len; p.desc = desc;
    p.out = out; p.sgram = sgram; p.tb = host_tables();
    p.rir_unit_stride = us; p.rir_chan_stride = cs; p.rir_elem_stride = es; p.rir_cap = cap;
    p.n_valid = n_valid; p.out_len = out_len;
    p.n_frames = 1 + out_len / ssk::kHop;
    p.t4 = (p.n_frames + 3) / 4;
    p.pad_mode = pad_mode;
    p.hspec = nullptr; p.h_blocks = 0; p.xcd_map = 0; p.stash = nullptr; p.stash_nbh = 0; p.stash_terms = 0; p.n_terms = 1;
    apply_bucket2(p);
    p.fade_len = 0;
    p.nb_y = 1;
    if (n_valid > ssk::kB || cap > ssk::kB || (fuse && (out_len > ssk::kB || p.t4 > 26))) return -1;
    ssk::UnitTab<true> ut;
    if (use_tab) {
        if (n_units > ssk::kTabUnits) return -2;
        for (int k = 0; k < n_units; ++k) {                 // launch slot k renders unit n - 1 - k (the library deals the units
            const int i = n_units - 1 - k;                  // out sorted by window spectrum: word 2 = where the results go)
            const int* d = desc + 8 * i;
            const bool ok = d[0] >= 0 && d[2] <= 0 && d[2] + d[3] > 0;
            ut.tab[ssk::kTabWords * k] = ok ? d[0] : -1;
            ut.tab[ssk::kTabWords * k + 1] = ok ? d[1] - d[2] : 0;
            ut.tab[ssk::kTabWords * k + 2] = i;
        }
    }
    gridDim = dim3{(unsigned)(2 * n_units), 1, 1};
    for (int b = 0; b < 2 * n_units; ++b) {
        blockIdx = dim3{(unsigned)b, 0, 0};
        int rc = run_block(ssk::kT32, [&] {
            if (fuse) { if (use_tab) ssk::k_conv32<true, true>(p, ut); else ssk::k_conv32<true, false>(p); }
            else { if (use_tab) ssk::k_conv32<false, true>(p, ut); else ssk::k_conv32<false, false>(p); }
        });
        if (rc) return rc;
    }
    return 0;
}

// spectral RIR bank: block spectra of every (entry, ear), like ss_rir_spectra_f32 (k_source_windows with scale 1)
int hs_rir_spectra(const float* rir, float* hspec, int n_entries, long long us, int cs, int cap) {
    const int hb = (cap + ssk::kB - 1) / ssk::kB;
    std::vector<int> desc;
    for (int r = 0; r < n_entries; ++r)
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < hb; ++i) {
                const int left = cap - i * ssk::kB;
                desc.push_back(static_cast<int>(r * us + (long long)c * cs + (long long)i * ssk::kB));
                desc.push_back(left < ssk::kB ? left : ssk::kB);
                desc.push_back(0);
                desc.push_back(0);
            }
    ssk::SrcParams p;
    p.src = rir; p.desc = desc.data(); p.spec = reinterpret_cast<ssk::f32x4*>(hspec); p.tb = host_tables();
    p.desc_stride = 4; p.scale = 1.0f;
    const int n_windows = static_cast<int>(desc.size() / 4);
    gridDim = dim3{(unsigned)n_windows, 1, 1};
    for (int w = 0; w < n_windows; ++w) {
        blockIdx = dim3{(unsigned)w, 0, 0};
        int rc = run_block(ssk::kT, [&] { ssk::k_source_windows(p); });
        if (rc) return rc;
    }
    return 0;
}

int hs_conv_spec(int fuse, int simple, const float* spec, const float* hspec, const int* rir_len, const int* desc,
                 float* out, float* sgram, int n_units, int h_blocks, int n_valid, int out_len, int pad_mode, int persist) {
    ssk::ConvParams p;
    p.spec = reinterpret_cast<const ssk::f32x4*>(spec); p.rir = nullptr; p.rir_len = rir_len; p.desc = desc;
    p.out = out; p.sgram = sgram; p.tb = host_tables();
    p.rir_unit_stride = 0; p.rir_chan_stride = 0; p.rir_elem_stride = 1; p.rir_cap = 0;
    p.n_valid = n_valid; p.out_len = out_len;
    p.n_frames = 1 + out_len / ssk::kHop;
    p.t4 = (p.n_frames + 3) / 4;
    p.pad_mode = pad_mode;
    p.fade_len = 0;
    p.hspec = reinterpret_cast<const ssk::f32x4*>(hspec);
    p.h_blocks = h_blocks; p.xcd_map = 0; p.stash = nullptr; p.stash_nbh = 0; p.stash_terms = 0; p.n_terms = 2; p.parts_log2 = 0;
    apply_bucket2(p);
    const int nb_y = n_valid == 0 ? 1 : (n_valid + ssk::kB - 1) / ssk::kB;
    // WIDE: a row longer than one block of which only block 0 is rendered (the library's wide_one_block_ok)
    const bool wide = fuse && out_len > ssk::kB && n_valid <= ssk::kB && ssk::live_blocks(n_valid, out_len, p.t4) <= 26;
    if (wide && simple) return -3;
    if (fuse && !wide && (nb_y != 1 || out_len > ssk::kB || p.t4 > 26)) return -1;
    if (simple && (nb_y != 1 || h_blocks != 1)) return -2;
    p.nb_y = nb_y;
    if (persist > 0) {                                  // k_conv_spec_rows: `persist` workgroups walk the units
        if (fuse || !simple) return -3;
        gridDim = dim3{(unsigned)persist, 1, 1};
        for (int b = 0; b < persist && b < n_units; ++b) {
            blockIdx = dim3{(unsigned)b, 0, 0};
            int rc = run_block(ssk::kT, [&] { ssk::k_conv_spec_rows(p, 2 * n_units); });
            if (rc) return rc;
        }
        return 0;
    }
    p.parts_log2 = fuse ? g_parts_log2 : 0;
    g_parts_log2 = 0;
    gridDim = dim3{(unsigned)((2 * n_units * nb_y) << p.parts_log2), 1, 1};
    for (int b = 0; b < (2 * n_units * nb_y) << p.parts_log2; ++b) {
        {
            blockIdx = dim3{(unsigned)b, 0, 0};
            int rc = run_block(ssk::kT, [&] {
                if (fuse) { if (simple) ssk::k_conv_spec<true, true>(p); else ssk::k_conv_spec<true, false>(p); }
                else { if (simple) ssk::k_conv_spec<false, true>(p); else ssk::k_conv_spec<false, false>(p); }
            });
            if (rc) return rc;
        }
    }
    return 0;
}

// k_obs_rows: `wgs` persistent workgroups walk the (unit, ear) rows; hspec != nullptr selects the spectral-bank variant
int hs_obs_rows(const float* spec, const float* rir, const float* hspec, const int* rir_len, const int* desc, float* out,
                float* sgram, int n_units, long long us, int cs, int es, int cap, int h_blocks, int n_valid, int out_len,
                int pad_mode, int wgs, int no_distractor, int use_stash, int crossfade) {
    if (out_len <= ssk::kB || out_len > 3 * ssk::kB || !sgram) return -1;
    ssk::ConvParams p;
    p.spec = reinterpret_cast<const ssk::f32x4*>(spec); p.rir = rir; p.rir_len = rir_len; p.desc = desc;
    p.out = out; p.sgram = sgram; p.tb = host_tables();
    p.rir_unit_stride = us; p.rir_chan_stride = cs; p.rir_elem_stride = es; p.rir_cap = cap;
    p.n_valid = n_valid; p.out_len = out_len;
    p.n_frames = 1 + out_len / ssk::kHop;
    p.t4 = (p.n_frames + 3) / 4;
    p.pad_mode = pad_mode;
    p.fade_len = static_cast<int>(0.05 * out_len);
    if (crossfade && (hspec || no_distractor || p.fade_len > 2 * ssk::kPrevPairs - 2)) return -2;
    p.hspec = reinterpret_cast<const ssk::f32x4*>(hspec); p.h_blocks = h_blocks; p.xcd_map = wgs >= 8;
    p.nb_y = n_valid == 0 ? 0 : (n_valid + ssk::kB - 1) / ssk::kB;
    p.n_terms = no_distractor ? 1 : 2;
    apply_bucket2(p);
    p.parts_log2 = g_parts_log2;                        // split rows: one workgroup per (row, part), whatever `wgs` says
    g_parts_log2 = 0;
    const int n_rows = 2 * n_units;
    if (g_obs_blocks) {
        const bool reversed = g_obs_blocks == 2;        // consumers BEFORE producers: every hand-off wait runs out (poisoned frames)
        g_obs_blocks = 0;
        if (crossfade || n_valid != out_len) return -3;
        const int nb = (out_len + ssk::kB - 1) / ssk::kB, grid_b = (n_rows * nb) << p.parts_log2;
        p.nb_y = nb;
        p.xcd_map = 0;                                  // workgroups run in blockIdx order here: (row, j - 1) before (row, j)
        p.stash = nullptr; p.stash_nbh = 0; p.stash_terms = 0;
        std::vector<float> tails(static_cast<size_t>(n_rows) * 2 * ssk::kTailFloats, 12345.0f);
        std::vector<int> fl(static_cast<size_t>(n_rows) * 2, 0);
        gridDim = dim3{(unsigned)grid_b, 1, 1};
        for (int bb = 0; bb < grid_b; ++bb) {
            const int b = reversed ? grid_b - 1 - bb : bb;
            blockIdx = dim3{(unsigned)b, 0, 0};
            int rc = run_block(ssk::kT, [&] {
                if (hspec) ssk::k_obs_blocks<true>(p, n_rows, tails.data(), fl.data(), 7);
                else ssk::k_obs_blocks<false>(p, n_rows, tails.data(), fl.data(), 7);
            });
            if (rc) return rc;
        }
        for (int r = 0; r < n_rows; ++r)                // every hand-off of a non-silent row was released exactly once
            for (int j = 0; j + 1 < nb; ++j)
                if (fl[static_cast<size_t>(r) * (nb - 1) + j] != 7 && fl[static_cast<size_t>(r) * (nb - 1) + j] != 0) return -4;
        return 0;
    }
    const int grid = p.parts_log2 ? (n_rows << p.parts_log2) : (wgs < n_rows ? wgs : n_rows);
    std::vector<float> stash;
    p.stash = nullptr; p.stash_nbh = 0; p.stash_terms = 0;
    (void)use_stash;                                    // (the time-domain path always has its stash)
    if (!hspec) {
        p.stash_nbh = (cap + ssk::kB - 1) / ssk::kB;
        if (p.n_buckets > 1 && p.bk[0].h_blocks > p.stash_nbh) p.stash_nbh = p.bk[0].h_blocks;
        p.stash_terms = p.n_terms;
        stash.assign(static_cast<size_t>(grid) * p.stash_terms * p.stash_nbh * 2 * ssk::kSpecComplex, 12345.0f);
        p.stash = reinterpret_cast<ssk::f32x4*>(stash.data());
    }
    gridDim = dim3{(unsigned)grid, 1, 1};
    for (int b = 0; b < grid; ++b) {
        blockIdx = dim3{(unsigned)b, 0, 0};
        int rc = run_block(ssk::kT, [&] {
            if (hspec) ssk::k_obs_rows<true>(p, n_rows);
            else if (crossfade) ssk::k_obs_rows<false, true>(p, n_rows);
            else ssk::k_obs_rows<false>(p, n_rows);
        });
        if (rc) return rc;
    }
    return 0;
}

int hs_spectrogram(const float* x, float* out, int n_units, int len, int pad_mode, int gpw) {
    ssk::SpecParams p;
    p.x = x; p.out = out; p.tb = host_tables();
    p.len = len; p.n_frames = 1 + len / ssk::kHop; p.t4 = (p.n_frames + 3) / 4; p.pad_mode = pad_mode;
    p.live = g_spec_n_valid >= 0 ? ssk::live_blocks(g_spec_n_valid, len, p.t4) : p.t4;
    g_spec_n_valid = -1;
    const int groups = (p.t4 + 3) / 4;
    p.gpw = gpw < 1 ? 1 : gpw > groups ? groups : gpw;
    const int chunks = (groups + p.gpw - 1) / p.gpw;
    gridDim = dim3{(unsigned)(n_units * chunks), 1, 1};
    for (int b = 0; b < n_units * chunks; ++b) {
        blockIdx = dim3{(unsigned)b, 0, 0};
        int rc = run_block(512, [&] { ssk::k_spectrogram(p); });
        if (rc) return rc;
    }
    return 0;
}

int hs_logmel(const float* x, float* out, int n_units, int len, int pad_mode, const int* start, const float* w,
              int n_mels, int max_len, float eps, int gpw) {
    ssk::MelParams p;
    p.x = x; p.out = out; p.tb = host_tables(); p.start = start; p.w = w;
    p.len = len; p.n_frames = 1 + len / ssk::kHop; p.pad_mode = pad_mode;
    p.n_mels = n_mels; p.max_len = max_len; p.eps = eps;
    if (n_mels > ssk::kMelMaxBands || max_len > ssk::kMelMaxLen || (max_len & 3) || n_mels * max_len > ssk::kMelTableFloats) return -2;
    const int groups = (p.n_frames + ssk::kSegFrames - 1) / ssk::kSegFrames;
    p.gpw = gpw < 1 ? 1 : gpw > groups ? groups : gpw;
    const int chunks = (groups + p.gpw - 1) / p.gpw;
    gridDim = dim3{(unsigned)(n_units * chunks), 1, 1};
    for (int b = 0; b < n_units * chunks; ++b) {
        blockIdx = dim3{(unsigned)b, 0, 0};
        int rc = run_block(512, [&] { ssk::k_logmel(p); });
        if (rc) return rc;
    }
    return 0;
}

int hs_gccphat(const float* x, float* out, int n_units, int len, int pad_mode, int max_lag, float eps, int gpw) {
    ssk::GccParams p;
    p.x = x; p.out = out; p.tb = host_tables();
    p.len = len; p.n_frames = 1 + len / ssk::kHop; p.pad_mode = pad_mode; p.max_lag = max_lag; p.eps = eps;
    if (max_lag < 1 || max_lag > ssk::kGccMaxLag) return -2;
    const int groups = (p.n_frames + ssk::kSegFrames - 1) / ssk::kSegFrames;
    p.gpw = gpw < 1 ? 1 : gpw > groups ? groups : gpw;
    const int chunks = (groups + p.gpw - 1) / p.gpw;
    gridDim = dim3{(unsigned)(n_units * chunks), 1, 1};
    for (int b = 0; b < n_units * chunks; ++b) {
        blockIdx = dim3{(unsigned)b, 0, 0};
        int rc = run_block(256, [&] { ssk::k_gccphat(p); });
        if (rc) return rc;
    }
    return 0;
}

int hs_features(const float* x, int n_units, int len, int pad_mode, float* sgram, float* mel, const int* start, const float* w,
                int n_mels, int max_len, float mel_eps, float* gcc, int max_lag, float gcc_eps, int gpw) {
    ssk::FeatParams p;
    p.x = x; p.sgram = sgram; p.mel = mel; p.gcc = gcc; p.tb = host_tables(); p.mel_start = start; p.mel_w = w;
    p.len = len; p.n_frames = 1 + len / ssk::kHop; p.t4 = (p.n_frames + 3) / 4; p.pad_mode = pad_mode;
    p.n_mels = mel ? n_mels : 0; p.max_len = mel ? max_len : 4; p.max_lag = gcc ? max_lag : 1;
    p.mel_eps = mel_eps; p.gcc_eps = gcc_eps;
    if (mel && (n_mels > ssk::kFeatMaxMels || max_len > ssk::kFeatMaxLen || (max_len & 3) || n_mels * max_len > ssk::kFeatMelTable)) return -2;
    if (gcc && (max_lag < 1 || max_lag > ssk::kGccMaxLag)) return -2;
    const int groups = (p.n_frames + ssk::kSegFrames - 1) / ssk::kSegFrames;
    p.n_units = n_units;
    const int tasks = n_units * groups;                 // gpw: rounds per workgroup wanted -> grid = ceil(tasks / gpw)
    const int wgs = (tasks + (gpw < 1 ? 1 : gpw) - 1) / (gpw < 1 ? 1 : gpw);
    gridDim = dim3{(unsigned)wgs, 1, 1};
    for (int b = 0; b < wgs; ++b) {
        blockIdx = dim3{(unsigned)b, 0, 0};
        int rc = run_block(256, [&] {
            const int which = (mel ? 1 : 0) | (sgram ? 2 : 0) | (gcc ? 4 : 0);
            switch (which) {
                case 1: ssk::k_features<true, false, false>(p); break;
                case 2: ssk::k_features<false, true, false>(p); break;
                case 3: ssk::k_features<true, true, false>(p); break;
                case 4: ssk::k_features<false, false, true>(p); break;
                case 5: ssk::k_features<true, false, true>(p); break;
                case 6: ssk::k_features<false, true, true>(p); break;
                default: ssk::k_features<true, true, true>(p); break;
            }
        });
        if (rc) return rc;
    }
    return 0;
}

int hs_intensity(const float* x, float* out, int n_units, int len, int num_frame) {
    ssk::IntensityParams p;
    p.x = x; p.out = out; p.len = len; p.num_frame = num_frame;
    gridDim = dim3{(unsigned)n_units, 1, 1};
    for (int b = 0; b < n_units; ++b) {
        blockIdx = dim3{(unsigned)b, 0, 0};
        int rc = run_block(256, [&] { ssk::k_intensity(p); });
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"

// ss_hip.hip — host side of libss_hip.so: table construction and kernel launches (gfx950 only).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <iterator>
#include <map>
#include <mutex>
#include <utility>
#include <new>
#include <string>
#include <vector>

#include "../../include/ss_hip.h"
#include "ss_context.hpp"
#include "ss_kernels.hpp"
#include "ss_kernels32.hpp"
#include "ss_features.hpp"
#include "ss_tables.hpp"
#include "ss_wavio.hpp"

namespace {

constexpr int kMaxDevices = 64;
struct DeviceTables {
    bool ready = false;
    ssk::Tables tb{};
    int n_cus = 0;
};
DeviceTables g_tables[kMaxDevices];
std::mutex g_mu;

inline int hip_err(hipError_t e) { return e == hipSuccess ? 0 : -static_cast<int>(e); }

// A/B switches for benchmarking (scripts/gpu_ab_*.sh) exist only in -DSS_AB builds: the product library reads no
// environment variables ("benchmark knobs inside a shipped .so")
#if defined(SS_AB)
inline bool ab_flag(const char* name) { return std::getenv(name) != nullptr; }
inline int ab_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }
#else
constexpr bool ab_flag(const char*) { return false; }
constexpr int ab_int(const char*, int dflt) { return dflt; }
#endif

// Host time of one ss_ctx_observe call by segment (A/B builds only; ss_ab_host_profile reads and clears the sums)
#if defined(SS_AB)
double g_prof_ns[8] = {0};
long long g_prof_calls = 0;
struct ProfClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(int seg) {
        const auto n = std::chrono::steady_clock::now();
        g_prof_ns[seg] += std::chrono::duration<double, std::nano>(n - t).count();
        t = n;
    }
};
#define SS_PROF_BEGIN() ProfClock prof_clock
#define SS_PROF_MARK(seg) prof_clock.mark(seg)
#define SS_PROF_CALL() (++g_prof_calls)
#else
#define SS_PROF_BEGIN() ((void)0)
#define SS_PROF_MARK(seg) ((void)0)
#define SS_PROF_CALL() ((void)0)
#endif

int get_tables(ssk::Tables* out, int* n_cus = nullptr) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_err(e);
    if (dev < 0 || dev >= kMaxDevices) return SS_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceTables& d = g_tables[dev];
    if (!d.ready) {
        const std::vector<float> host = ssk_host::build_tables();
        float* dev_buf = nullptr;
        e = hipMalloc(&dev_buf, host.size() * sizeof(float));
        if (e != hipSuccess) return hip_err(e);
        e = hipMemcpy(dev_buf, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(dev_buf); return hip_err(e); }
        d.tb.twM = reinterpret_cast<const ssk::c32*>(dev_buf + ssk_host::kTwMOff);
        d.tb.twItem = reinterpret_cast<const ssk::c32*>(dev_buf + ssk_host::kTwItemOff);
        d.tb.tw512 = reinterpret_cast<const ssk::c32*>(dev_buf + ssk_host::kTw512Off);
        d.tb.win = dev_buf + ssk_host::kWinOff;
        d.tb.twG = reinterpret_cast<const ssk::c32*>(dev_buf + ssk_host::kTwGOff);
        d.tb.twP2 = reinterpret_cast<const ssk::c32*>(dev_buf + ssk_host::kTwP2Off);
        int cus = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);   // slow call: once per device
        if (e != hipSuccess) { (void)hipFree(dev_buf); return hip_err(e); }
        d.n_cus = cus > 0 ? cus : 1;
        d.ready = true;
    }
    *out = d.tb;
    if (n_cus) *n_cus = d.n_cus;
    return 0;
}

inline int n_frames_of(int len) { return 1 + len / ssk::kHop; }
inline int t4_of(int len) { return (n_frames_of(len) + ssk::kPool - 1) / ssk::kPool; }

// The context API knows the unit descriptors of a step on the HOST (it has just planned them): for the duration of its
// launch call it leaves them here, and loop-free launches of <= kTabUnits units then carry {index, slot} per unit in the
// kernel arguments (ConvParams::tab).  Callers of the stateless entry points hand over device pointers only: nullptr.
thread_local const int* g_host_desc = nullptr;
// ... and, in its overlap mode, the number of launches it keeps in flight (its lanes): a launch that spreads its rows over
// idle CUs (parts_log2_for) then leaves the other lanes' launches their share of the chip.  1 for everybody else.
thread_local int g_launch_share = 1;

inline bool fill_unit_tab(ssk::UnitTab<true>& ut, const int* host_desc, int n_units) {
    if (!host_desc || n_units > ssk::kTabUnits) return false;
    // Units are dealt to the launch slots SORTED by their window spectrum (stable; silent units last): row_slot() gives every
    // XCD a contiguous range of slots, so the rows that read one 128-KiB spectrum - 128 envs draw from 102 sounds, ~73 distinct
    // per step - meet in ONE L2 instead of pulling it through several (r4 PMC: TCC hit rate 0.55, 1.9 x the algorithmic bytes).
    // Word 2 of an entry = the unit the row's results belong to, so the caller's order of outputs is untouched.
    static const bool no_sort = ab_flag("SS_HIP_NO_SORT");            // (A/B builds only)
    unsigned key[ssk::kTabUnits];                                     // (slot << 8) | unit: 32-bit keys, ~1 us for 128 units
    // (steps of >= 64 units only: with fewer, few rows share a spectrum and the ~2 us of host time per call - measured:
    // 13.9 vs 11.7 us at 128 units - are the bottleneck of a small step, not its bytes)
    bool sortable = !no_sort && n_units >= 64;
    for (int i = 0; i < n_units; ++i) {
        const int* d = host_desc + 8 * i;
        const bool ok = d[0] >= 0 && d[2] <= 0 && d[2] + d[3] > 0;          // window m = 0 of the unit's key is stored
        const int slot = ok ? d[1] - d[2] : 0xffffff;
        if (slot < 0 || slot > 0xffffff) sortable = false;
        key[i] = (static_cast<unsigned>(slot & 0xffffff) << 8) | static_cast<unsigned>(i);
    }
    if (sortable) std::sort(key, key + n_units);
    for (int k = 0; k < n_units; ++k) {
        const int i = sortable ? static_cast<int>(key[k] & 0xffu) : k;
        const int* d = host_desc + 8 * i;
        const bool ok = d[0] >= 0 && d[2] <= 0 && d[2] + d[3] > 0;
        ut.tab[ssk::kTabWords * k] = ok ? d[0] : -1;
        ut.tab[ssk::kTabWords * k + 1] = ok ? d[1] - d[2] : 0;
        ut.tab[ssk::kTabWords * k + 2] = i;
    }
    return true;
}

// Rows longer than one block of which only block 0 is rendered (SS2.0 steps at 44.1 kHz: 0.25 s of a 1-s row) and whose live
// pooled blocks all lie inside that block: the fused LOOP kernel serves them in one launch (k_conv<..., WIDE>: block spectra
// accumulated in registers, the spectrogram's dead columns written as zeros) - with or without a waveform buffer.
inline bool wide_one_block_ok(int out_len, int n_valid, int flags, bool spectral = false) {
    static const bool off = ab_flag("SS_HIP_NO_WIDE");        // (A/B builds only: -DSS_AB)
    if (off || spectral || out_len <= ssk::kB || n_valid < 0 || n_valid > ssk::kB) return false;
    if ((flags & SS_FLAG_CROSSFADE) &&
        (static_cast<int>(0.05 * out_len) < 1 || static_cast<int>(0.05 * out_len) > 2 * ssk::kPrevPairs - 2)) return false;
    return ssk::live_blocks(n_valid, out_len, t4_of(out_len)) <= 26;
}

// Small steps (fewer rows than CUs): a fused one-block row is rendered by 2^k workgroups on as many CUs, each running the
// row's convolution and its share of the pooled STFT blocks (ConvParams::parts_log2) - as long as the grid still fits the
// chip in one round of one workgroup per CU.  26 pooled blocks: k <= 3 (shares of 13 / 7 / 4 blocks).
inline int parts_log2_for(int n_rows, int n_cus) {
    static const int forced = ab_int("SS_HIP_PARTS_LOG2", -1);        // (A/B builds only: -DSS_AB)
    if (forced >= 0) return forced < 3 ? forced : 3;
    const int cus = n_cus / (g_launch_share > 1 ? g_launch_share : 1);       // overlap mode: the lanes share the chip
    int k = 0;
    while (k < 3 && (n_rows << (k + 1)) <= cus) ++k;
    return k;
}

template <bool FUSE>
int launch_conv(ssk::ConvParams p, int n_units, int nb_y, int flags, int n_cus, hipStream_t st, bool wide = false) {
    if (nb_y < 1 || nb_y > 3 || (FUSE && nb_y != 1)) return SS_EINVAL;
    p.nb_y = nb_y;
    p.parts_log2 = FUSE && !wide ? parts_log2_for(2 * n_units, n_cus) : 0;
    if constexpr (FUSE) {
        if (wide) {
            if ((flags & SS_FLAG_CROSSFADE) && (p.fade_len < 1 || p.fade_len > 2 * ssk::kPrevPairs - 2)) return SS_EINVAL;
            const dim3 grid(2 * n_units, 1), block(ssk::kT);
            if (flags & SS_FLAG_CROSSFADE)
                hipLaunchKernelGGL((ssk::k_conv<true, false, true, false, true>), grid, block, 0, st, p, ssk::UnitTab<false>());
            else
                hipLaunchKernelGGL((ssk::k_conv<true, false, false, false, true>), grid, block, 0, st, p, ssk::UnitTab<false>());
            return hip_err(hipGetLastError());
        }
    }
    // the loop-free kernels address bucket 0 only: a bucketed bank qualifies when the caller promises that every index
    // of the launch lies there (SS_FLAG_FIRST_BUCKET; the context's planner works it out per step)
    const bool bucket0 = p.n_buckets == 1 || (flags & SS_FLAG_FIRST_BUCKET);
    const bool simple = (flags & SS_FLAG_NO_DISTRACTOR) && !(flags & SS_FLAG_CROSSFADE) && nb_y == 1 &&
                        p.rir_cap <= ssk::kB && bucket0;
    if ((flags & SS_FLAG_CROSSFADE) && (p.fade_len < 1 || p.fade_len > 2 * ssk::kPrevPairs - 2)) return SS_EINVAL;
    const dim3 grid((2 * n_units) << p.parts_log2, nb_y), block(ssk::kT);
    // more rows than CUs: persistent workgroups that prefetch the next row's RIR under the current row's FFT passes
    const bool planar = p.rir_elem_stride == 1 && !(p.rir_cap & 1) && !(reinterpret_cast<size_t>(p.rir) & 7) &&
                        !(p.rir_unit_stride & 1) && !(p.rir_chan_stride & 1) && p.rir_cap >= 2;
    static const bool no_rows = ab_flag("SS_HIP_NO_ROW_KERNEL");
    if constexpr (!FUSE) {              // the fused kernel gains nothing from it (measured), see k_conv_rows
        if (simple && planar && !no_rows && 2 * n_units > n_cus) {
            hipLaunchKernelGGL(ssk::k_conv_rows, dim3(n_cus), block, 0, st, p, 2 * n_units);
            return hip_err(hipGetLastError());
        }
    }
    if (flags & SS_FLAG_CROSSFADE) hipLaunchKernelGGL((ssk::k_conv<FUSE, false, true>), grid, block, 0, st, p, ssk::UnitTab<false>());
    else if (simple) {
        ssk::UnitTab<true> ut;
        if (fill_unit_tab(ut, g_host_desc, n_units)) hipLaunchKernelGGL((ssk::k_conv<FUSE, true, false, true>), grid, block, 0, st, p, ut);
        else hipLaunchKernelGGL((ssk::k_conv<FUSE, true>), grid, block, 0, st, p, ssk::UnitTab<false>());
    }
    else hipLaunchKernelGGL((ssk::k_conv<FUSE, false>), grid, block, 0, st, p, ssk::UnitTab<false>());
    return hip_err(hipGetLastError());
}

// ---- k_obs_rows: fused observation for rows longer than one partition block -------------------------------------
// The kernel's workgroups keep the block spectra of the row they are rendering in a private stash (see k_obs_rows).
// Launches on ONE stream run one after the other, so the stash is owned by (device, stream): two streams never share
// one, whatever their launches overlap with.  Grown on demand (first long-row launch of a stream), never shrunk.
struct StashBuf { float* ptr = nullptr; size_t bytes = 0; };
std::map<std::pair<int, hipStream_t>, StashBuf> g_stash;

int get_stash(hipStream_t st, size_t bytes, float** out) {
    // one handle for a different stream on every thread: two concurrent launches would share one stash
    if (st == hipStreamPerThread) return SS_EINVAL;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_err(e);
    std::lock_guard<std::mutex> lk(g_mu);
    StashBuf& b = g_stash[std::make_pair(dev, st)];
    if (b.bytes < bytes) {
        if (b.ptr) {
            e = hipStreamSynchronize(st);                       // earlier launches of this stream still use the old one
            if (e != hipSuccess) return hip_err(e);
            (void)hipFree(b.ptr);
            b.ptr = nullptr; b.bytes = 0;
        }
        e = hipMalloc(reinterpret_cast<void**>(&b.ptr), bytes);
        if (e != hipSuccess) { b.ptr = nullptr; return hip_err(e); }
        b.bytes = bytes;
    }
    *out = b.ptr;
    return 0;
}

// k_obs_blocks' hand-off area per (device, stream): [kSyncRows][2] tails of kTailFloats floats + as many flag words, zeroed once;
// `epoch` counts the launches that used it (a flag holds the epoch of the launch that set it: nothing is reset in between -
// which is also why such a launch cannot be replayed from a captured graph: the library never captures)
constexpr int kSyncRows = 512;                                  // (unit, ear) rows a k_obs_blocks launch may have: more than CUs / 2
struct SyncBuf { float* tails = nullptr; int* flags = nullptr; int epoch = 0; };
std::map<std::pair<int, hipStream_t>, SyncBuf> g_sync;

int get_block_sync(hipStream_t st, float** tails, int** flags, int* epoch) {
    if (st == hipStreamPerThread) return SS_EINVAL;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_err(e);
    std::lock_guard<std::mutex> lk(g_mu);
    SyncBuf& b = g_sync[std::make_pair(dev, st)];
    if (!b.tails) {
        const size_t n_hand = static_cast<size_t>(kSyncRows) * 2;
        const size_t bytes = n_hand * ssk::kTailFloats * sizeof(float) + n_hand * sizeof(int);
        void* ptr = nullptr;
        e = hipMalloc(&ptr, bytes);
        if (e != hipSuccess) return hip_err(e);
        e = hipMemsetAsync(ptr, 0, bytes, st);                  // (stream-ordered in front of the first launch that reads it)
        if (e != hipSuccess) { (void)hipFree(ptr); return hip_err(e); }
        b.tails = static_cast<float*>(ptr);
        b.flags = reinterpret_cast<int*>(b.tails + n_hand * ssk::kTailFloats);
        b.epoch = 0;
    }
    b.epoch = b.epoch == 0x7fffffff ? 1 : b.epoch + 1;
    *tails = b.tails; *flags = b.flags; *epoch = b.epoch;
    return 0;
}

// a stream is about to be destroyed (the context's overlap lanes): its stash goes with it (~96 MiB per lane
// leaked by every create / destroy of a context with overlap); the caller has synchronised the stream
void drop_stash(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto sy = g_sync.find(std::make_pair(dev, st));
    if (sy != g_sync.end()) {
        if (sy->second.tails) (void)hipFree(sy->second.tails);
        g_sync.erase(sy);
    }
    auto it = g_stash.find(std::make_pair(dev, st));
    if (it == g_stash.end()) return;
    if (it->second.ptr) (void)hipFree(it->second.ptr);
    g_stash.erase(it);
}

// should k_obs_rows serve this shape?  (rows of 2 or 3 blocks, stash mask of 32 bits.)  Rows with ONE rendered block
// (n_valid <= kB: SS2.0 steps at 44.1 kHz) and cross-faded rows: only when the caller has no waveform buffer.  With one,
// the loop kernel (block spectra accumulated in registers: every one is used once, the rows kernel's stash is pure overhead
// there) + k_spectrogram (told where the zeros of a short step begin) is faster - per 128 units at 44.1 kHz: plain steps
// 61.6 vs 72.5 us (66.3 without the waveform written), cross-faded 92.7 vs 117.8 us (profiles/r3/NOTES.md section 8)
inline bool obs_rows_ok(int out_len, int n_valid, int nbh_max, int flags, bool spectral = false, bool have_waveform_buffer = false) {
    if (have_waveform_buffer && ((flags & SS_FLAG_CROSSFADE) || n_valid <= ssk::kB)) return false;
    if ((flags & SS_FLAG_CROSSFADE) &&
        (spectral || static_cast<int>(0.05 * out_len) < 1 || static_cast<int>(0.05 * out_len) > 2 * ssk::kPrevPairs - 2)) return false;
    return out_len > ssk::kB && out_len <= 3 * ssk::kB && n_valid <= out_len && nbh_max >= 1 && nbh_max <= 16;
}

// Small steps of rows longer than one block (the reference's Replica arrangement: 5 envs per GPU at 44.1 kHz): one workgroup
// per OUTPUT BLOCK of a row instead of one per row (k_obs_blocks) - while the whole grid fits the launch's share of the chip at
// one workgroup per CU, which is also what makes the kernel's inter-workgroup hand-off safe.  Spare CUs go to parts (the STFT
// phase of a block split 2 / 4 / 8 ways).  Returns 1 when the launch does not qualify (the caller falls through to k_obs_rows).
template <bool SPECTRAL>
int launch_obs_blocks(ssk::ConvParams p, int n_units, int flags, int n_cus, hipStream_t st) {
    static const bool off = ab_flag("SS_HIP_NO_OBS_BLOCKS");   // (A/B builds only)
    const int n_rows = 2 * n_units, nb = (p.out_len + ssk::kB - 1) / ssk::kB;
    const int budget = n_cus / (g_launch_share > 1 ? g_launch_share : 1);
    if (off || (flags & SS_FLAG_CROSSFADE) || p.n_valid != p.out_len || nb < 2 || nb > 3 || n_rows * nb > budget ||
        n_rows > kSyncRows || st == hipStreamPerThread)
        return 1;
    int k = 0;
    while (k < 3 && ((n_rows * nb) << (k + 1)) <= budget) ++k;
    p.parts_log2 = k;
    p.nb_y = nb;
    p.n_terms = (flags & SS_FLAG_NO_DISTRACTOR) ? 1 : 2;
    p.stash = nullptr; p.stash_nbh = 0; p.stash_terms = 0;
    float* tails = nullptr;
    int* fl = nullptr;
    int epoch = 0;
    int rc = get_block_sync(st, &tails, &fl, &epoch);
    if (rc) return rc;
    const int grid = (n_rows * nb) << k;
    hipLaunchKernelGGL((ssk::k_obs_blocks<SPECTRAL>), dim3(grid), dim3(ssk::kT), 0, st, p, n_rows, tails, fl, epoch);
    return hip_err(hipGetLastError());
}

template <bool SPECTRAL>
int launch_obs_rows(ssk::ConvParams p, int n_units, int flags, int n_cus, hipStream_t st) {
    const int n_rows = 2 * n_units;
    {
        const int rc = launch_obs_blocks<SPECTRAL>(p, n_units, flags, n_cus, st);
        if (rc != 1) return rc;
    }
    // small steps (the reference steps 5-10 envs per GPU at this rate): a row on 2 / 4 / 8 CUs, each rendering the row and
    // its share of every phase's pooled STFT blocks - only while every (row, part) still gets a workgroup of its own
    // Time-domain bank: every part also repeats the row's stash round trip (640 KiB per row), and the launch turns
    // memory-bound long before the CUs run out - same box, alternating, us per launch at 1 / 5 / 10 / 16 units with up to 8
    // parts per row: 58.7 / 62.2 / 69.5 / 75.4 against 70.8 / 71.1 / 72.1 / 72.6 with one (profiles/r5/kbench_parts_44k.txt) -
    // so the split stops at 96 workgroups there.  The spectral bank has no stash: -20 % up to 32 units (256 workgroups).
    p.parts_log2 = parts_log2_for(n_rows, SPECTRAL ? n_cus : std::min(n_cus, 96));
    const int grid = (n_rows << p.parts_log2) < n_cus ? (n_rows << p.parts_log2) : n_cus;
    p.nb_y = p.n_valid == 0 ? 0 : (p.n_valid + ssk::kB - 1) / ssk::kB;
    p.stash = nullptr;
    p.stash_nbh = 0;
    p.stash_terms = 0;
    p.n_terms = (flags & SS_FLAG_NO_DISTRACTOR) ? 1 : 2;
    const bool xfade = (flags & SS_FLAG_CROSSFADE) != 0;
    if (xfade && (SPECTRAL || p.n_terms != 2)) return SS_EINVAL;
    // Time-domain bank: the kernel transforms every RIR block once per row and keeps the spectra that are needed again
    // in a per-workgroup stash (k_obs_rows); 44.1 kHz: 2 x 128 KiB written and 3 x 128 KiB read back per row.
    if (!SPECTRAL) {
        int nbh_max = (p.rir_cap + ssk::kB - 1) / ssk::kB;
        for (int b = 0; b + 1 < p.n_buckets; ++b) nbh_max = std::max(nbh_max, (p.bk[b].cap + ssk::kB - 1) / ssk::kB);
        p.stash_terms = p.n_terms;                              // no distractor terms: half the stash
        p.stash_nbh = nbh_max;
        const size_t per_wg = static_cast<size_t>(p.stash_terms) * nbh_max * ssk::kSpecComplex * sizeof(ssk::c32);
        float* buf = nullptr;
        int rc = get_stash(st, per_wg * static_cast<size_t>(grid), &buf);
        if (rc) return rc;
        p.stash = reinterpret_cast<ssk::f32x4*>(buf);
    }
    if constexpr (!SPECTRAL) {
        if (xfade) {
            hipLaunchKernelGGL((ssk::k_obs_rows<false, true>), dim3(grid), dim3(ssk::kT), 0, st, p, n_rows);
            return hip_err(hipGetLastError());
        }
    }
    // one bank allocation (every launch but those of a length-bucketed store): the instantiation without the bucket descriptors
    if (p.n_buckets == 1) hipLaunchKernelGGL((ssk::k_obs_rows<SPECTRAL, false, false>), dim3(grid), dim3(ssk::kT), 0, st, p, n_rows);
    else hipLaunchKernelGGL((ssk::k_obs_rows<SPECTRAL>), dim3(grid), dim3(ssk::kT), 0, st, p, n_rows);
    return hip_err(hipGetLastError());
}

template <bool FUSE>
int launch_conv32(ssk::ConvParams& p, int n_units, hipStream_t st) {
    const dim3 grid(2 * n_units), block(ssk::kT32);
    ssk::UnitTab<true> ut;
    if (fill_unit_tab(ut, g_host_desc, n_units)) hipLaunchKernelGGL((ssk::k_conv32<FUSE, true>), grid, block, 0, st, p, ut);
    else hipLaunchKernelGGL((ssk::k_conv32<FUSE, false>), grid, block, 0, st, p, ssk::UnitTab<false>());
    return hip_err(hipGetLastError());
}

}  // namespace

extern "C" {

int ss_block_len(void) { return ssk::kB; }
int ss_spec_floats(void) { return 2 * ssk::kSpecComplex; }
int ss_version(void) { return 1; }

// Frees the per-(device, stream) scratch k_obs_rows keeps (see get_stash); synchronises the device first.
int ss_release_scratch(void) {
    int cur = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return hip_err(e);
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = 0;
    for (auto& kv : g_stash) {                                  // every entry's OWN device is synchronised before its free
        if (!kv.second.ptr) continue;
        e = hipSetDevice(kv.first.first);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { rc = hip_err(e); continue; }    // (left allocated: a launch may still be using it)
        (void)hipFree(kv.second.ptr);
        kv.second.ptr = nullptr;
        kv.second.bytes = 0;
    }
    for (auto it = g_stash.begin(); it != g_stash.end();) it = it->second.ptr ? std::next(it) : g_stash.erase(it);
    for (auto& kv : g_sync) {                                   // k_obs_blocks' hand-off areas likewise
        if (!kv.second.tails) continue;
        e = hipSetDevice(kv.first.first);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { rc = hip_err(e); continue; }
        (void)hipFree(kv.second.tails);
        kv.second.tails = nullptr;
    }
    for (auto it = g_sync.begin(); it != g_sync.end();) it = it->second.tails ? std::next(it) : g_sync.erase(it);
    (void)hipSetDevice(cur);
    return rc;
}

int ss_init(void) {
    ssk::Tables tb;
    return get_tables(&tb);
}

int ss_source_windows_f32(const float* src, const int* win_desc, float* spec_out, int n_windows, void* stream) {
    if (n_windows == 0) return 0;
    if (!src || !win_desc || !spec_out || n_windows < 0) return SS_EINVAL;
    ssk::SrcParams p;
    int rc = get_tables(&p.tb);
    if (rc) return rc;
    p.src = src;
    p.desc = win_desc;
    p.spec = reinterpret_cast<ssk::f32x4*>(spec_out);
    p.desc_stride = 4;
    p.scale = ssk::kWindowScale;
    hipLaunchKernelGGL(ssk::k_source_windows, dim3(n_windows), dim3(ssk::kT), 0,
                       static_cast<hipStream_t>(stream), p);
    return hip_err(hipGetLastError());
}

// window spectra scattered into a pool: desc rows {src_offset, src_len, start, wrap, pool slot}
static int launch_windows_scatter(const float* src, const int* win_desc5, float* pool, int n_windows, hipStream_t st) {
    ssk::SrcParams p;
    int rc = get_tables(&p.tb);
    if (rc) return rc;
    p.src = src;
    p.desc = win_desc5;
    p.spec = reinterpret_cast<ssk::f32x4*>(pool);
    p.desc_stride = 5;
    p.scale = ssk::kWindowScale;
    hipLaunchKernelGGL(ssk::k_source_windows, dim3(n_windows), dim3(ssk::kT), 0, st, p);
    return hip_err(hipGetLastError());
}

static int fill_conv(ssk::ConvParams& p, int* n_cus, const float* spec, const float* rir, const int* rir_len,
                     const int* unit_desc, long long us, int cs, int es, int cap, int n_valid, int out_len,
                     bool need_rir = true) {
    if (!spec || (need_rir && !rir) || !rir_len || !unit_desc) return SS_EINVAL;
    if (n_valid < 0 || out_len <= 0 || n_valid > out_len || n_valid > 3 * ssk::kB) return SS_EINVAL;
    if (es < 1 || cs < 0 || us < 0 || cap < 0) return SS_EINVAL;
    int rc = get_tables(&p.tb, n_cus);
    if (rc) return rc;
    p.spec = reinterpret_cast<const ssk::f32x4*>(spec);
    p.rir = rir;
    p.rir_len = rir_len;
    p.desc = unit_desc;
    p.out = nullptr;
    p.sgram = nullptr;
    p.rir_unit_stride = us;
    p.rir_chan_stride = cs;
    p.rir_elem_stride = es;
    p.rir_cap = cap;
    p.n_valid = n_valid;
    p.out_len = out_len;
    p.n_frames = n_frames_of(out_len);
    p.t4 = t4_of(out_len);
    p.pad_mode = 0;
    p.fade_len = static_cast<int>(0.05 * out_len);     // crossfade_samples = int(0.05 * sr), rows are 1 s (out_len == sr)
    p.hspec = nullptr;
    p.h_blocks = 0;
    // default on: the two ears of a unit (same window spectrum) share an XCD's L2; SS_HIP_XCD_MAP=0 is the A/B switch
    static const int xcd_map = ab_int("SS_HIP_XCD_MAP", 1);
    p.xcd_map = xcd_map;
    p.stash = nullptr;
    p.stash_nbh = 0;
    p.stash_terms = 0;
    p.n_terms = 2;
    p.parts_log2 = 0;
    p.n_buckets = 1;
    for (auto& b : p.bk) b = ssk::BankBucket{nullptr, nullptr, 0x7fffffff, 0, 0, 0};
#if defined(SS_LADDER)                                     // profiling builds only (scripts/gpu_ladder.sh compiles with -DSS_LADDER)
    static const int dbg = getenv("SS_HIP_DBG") ? atoi(getenv("SS_HIP_DBG")) : 0;
    p.dbg = dbg;
#endif
    return 0;
}

int ss_fftconv_binaural_f32(const float* spec, const float* rir, const int* rir_len, const int* unit_desc,
                            float* out, int n_units, long long rir_unit_stride, int rir_chan_stride,
                            int rir_elem_stride, int rir_cap, int n_valid, int out_len, int flags, void* stream) {
    if (n_units == 0) return 0;
    if (!out || n_units < 0) return SS_EINVAL;
    ssk::ConvParams p;
    int n_cus = 1;
    int rc = fill_conv(p, &n_cus, spec, rir, rir_len, unit_desc, rir_unit_stride, rir_chan_stride, rir_elem_stride,
                       rir_cap, n_valid, out_len);
    if (rc) return rc;
    p.out = out;
    const int nb_y = n_valid == 0 ? 1 : (n_valid + ssk::kB - 1) / ssk::kB;
    return launch_conv<false>(p, n_units, nb_y, flags, n_cus, static_cast<hipStream_t>(stream));
}

// n_valid: samples from n_valid on are KNOWN to be zero (rows this library rendered itself); len: nothing known
static int spectrogram_of_rows(const float* x, float* out, int n_units, int len, int n_valid, int pad_mode, void* stream);

int ss_spectrogram_f32(const float* x, float* out, int n_units, int len, int pad_mode, void* stream) {
    return spectrogram_of_rows(x, out, n_units, len, len, pad_mode, stream);
}

static int spectrogram_of_rows(const float* x, float* out, int n_units, int len, int n_valid, int pad_mode, void* stream) {
    if (n_units == 0) return 0;
    if (!x || !out || n_units < 0 || len < ssk::kNfft / 2 + 1) return SS_EINVAL;   // reflect pad needs len > 256
    if (pad_mode != SS_PAD_REFLECT && pad_mode != SS_PAD_CONSTANT) return SS_EINVAL;
    ssk::SpecParams p;
    int rc = get_tables(&p.tb);
    if (rc) return rc;
    p.x = x;
    p.out = out;
    p.len = len;
    p.n_frames = n_frames_of(len);
    p.t4 = t4_of(len);
    p.pad_mode = pad_mode;
    p.live = ssk::live_blocks(n_valid, len, p.t4);
    const int groups = (p.t4 + 3) / 4;                      // 4 pooled time blocks x 2 ears per round of a workgroup
    // large batches: one workgroup walks several groups (tables staged once, next segment prefetched under the math);
    // small batches keep one group per workgroup so that the launch still fills 256 CUs x 2 workgroups
    long long gpw = (long long)n_units * groups / 2048;
    p.gpw = gpw < 1 ? 1 : gpw > groups ? groups : (int)gpw;
    const int chunks = (groups + p.gpw - 1) / p.gpw;
    hipLaunchKernelGGL(ssk::k_spectrogram, dim3(n_units * chunks), dim3(512), 0,
                       static_cast<hipStream_t>(stream), p);
    return hip_err(hipGetLastError());
}

int ss_logmel_f32(const float* x, float* out, int n_units, int len, int pad_mode, const int* mel_start,
                  const float* mel_w, int n_mels, int max_len, float eps, void* stream) {
    if (n_units == 0) return 0;
    if (!x || !out || !mel_start || !mel_w || n_units < 0 || len < ssk::kNfft / 2 + 1) return SS_EINVAL;
    if (pad_mode != SS_PAD_REFLECT && pad_mode != SS_PAD_CONSTANT) return SS_EINVAL;
    if (n_mels < 1 || n_mels > ssk::kMelMaxBands || max_len < 4 || max_len > ssk::kMelMaxLen || (max_len & 3) ||
        n_mels * max_len > ssk::kMelTableFloats || !(eps > 0.f) || (reinterpret_cast<size_t>(mel_w) & 15))
        return SS_EINVAL;
    ssk::MelParams p;
    int rc = get_tables(&p.tb);
    if (rc) return rc;
    p.x = x;
    p.out = out;
    p.start = mel_start;
    p.w = mel_w;
    p.len = len;
    p.n_frames = n_frames_of(len);
    p.pad_mode = pad_mode;
    p.n_mels = n_mels;
    p.max_len = max_len;
    p.eps = eps;
    const int groups = (p.n_frames + ssk::kSegFrames - 1) / ssk::kSegFrames;
    long long gpw = (long long)n_units * groups / 1024;     // one 111 KB workgroup per CU: ~4 rounds of 256
    p.gpw = gpw < 1 ? 1 : gpw > groups ? groups : (int)gpw;
    const int chunks = (groups + p.gpw - 1) / p.gpw;
    hipLaunchKernelGGL(ssk::k_logmel, dim3(n_units * chunks), dim3(512), 0, static_cast<hipStream_t>(stream), p);
    return hip_err(hipGetLastError());
}

int ss_gccphat_f32(const float* x, float* out, int n_units, int len, int pad_mode, int max_lag, float eps,
                   void* stream) {
    if (n_units == 0) return 0;
    if (!x || !out || n_units < 0 || len < ssk::kNfft / 2 + 1) return SS_EINVAL;
    if (pad_mode != SS_PAD_REFLECT && pad_mode != SS_PAD_CONSTANT) return SS_EINVAL;
    if (max_lag < 1 || max_lag > ssk::kGccMaxLag || !(eps > 0.f)) return SS_EINVAL;
    ssk::GccParams p;
    int rc = get_tables(&p.tb);
    if (rc) return rc;
    p.x = x;
    p.out = out;
    p.len = len;
    p.n_frames = n_frames_of(len);
    p.pad_mode = pad_mode;
    p.max_lag = max_lag;
    p.eps = eps;
    const int groups = (p.n_frames + ssk::kSegFrames - 1) / ssk::kSegFrames;
    long long gpw = (long long)n_units * groups / 2048;
    p.gpw = gpw < 1 ? 1 : gpw > groups ? groups : (int)gpw;
    const int chunks = (groups + p.gpw - 1) / p.gpw;
    hipLaunchKernelGGL(ssk::k_gccphat, dim3(n_units * chunks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return hip_err(hipGetLastError());
}

int ss_audio_features_f32(const float* x, int n_units, int len, int pad_mode, float* spectrogram, float* logmel,
                          const int* mel_start, const float* mel_w, int n_mels, int max_len, float mel_eps, float* gccphat,
                          int max_lag, float gcc_eps, void* stream) {
    if (n_units == 0) return 0;
    if (!x || n_units < 0 || len < ssk::kNfft / 2 + 1 || (!spectrogram && !logmel && !gccphat)) return SS_EINVAL;
    if (pad_mode != SS_PAD_REFLECT && pad_mode != SS_PAD_CONSTANT) return SS_EINVAL;
    if (logmel && (!mel_start || !mel_w || n_mels < 1 || n_mels > ssk::kFeatMaxMels || max_len < 4 ||
                   max_len > ssk::kFeatMaxLen || (max_len & 3) || n_mels * max_len > ssk::kFeatMelTable || !(mel_eps > 0.f) ||
                   (reinterpret_cast<size_t>(mel_w) & 15)))
        return SS_EINVAL;
    if (gccphat && (max_lag < 1 || max_lag > ssk::kGccMaxLag || !(gcc_eps > 0.f))) return SS_EINVAL;
    ssk::FeatParams p;
    int rc = get_tables(&p.tb);
    if (rc) return rc;
    p.x = x; p.sgram = spectrogram; p.mel = logmel; p.gcc = gccphat;
    p.mel_start = mel_start; p.mel_w = mel_w;
    p.len = len; p.n_frames = n_frames_of(len); p.t4 = t4_of(len); p.pad_mode = pad_mode;
    p.n_mels = logmel ? n_mels : 0; p.max_len = logmel ? max_len : 4; p.max_lag = gccphat ? max_lag : 1;
    p.mel_eps = mel_eps; p.gcc_eps = gcc_eps;
    const int groups = (p.n_frames + ssk::kSegFrames - 1) / ssk::kSegFrames;
    // two ~80 KiB workgroups per CU: one wave of resident workgroups that share the (unit, group) rounds round-robin (tables
    // staged once per workgroup, the next round's segments prefetched under the current round's math)
    int n_cus = 256;
    { ssk::Tables unused; (void)get_tables(&unused, &n_cus); }
    p.n_units = n_units;
    const long long tasks = (long long)n_units * groups;
    const dim3 grid(static_cast<unsigned>(tasks < 2LL * n_cus ? tasks : 2LL * n_cus)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int which = (logmel ? 1 : 0) | (spectrogram ? 2 : 0) | (gccphat ? 4 : 0);
    switch (which) {
        case 1: hipLaunchKernelGGL((ssk::k_features<true, false, false>), grid, block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((ssk::k_features<false, true, false>), grid, block, 0, st, p); break;
        case 3: hipLaunchKernelGGL((ssk::k_features<true, true, false>), grid, block, 0, st, p); break;
        case 4: hipLaunchKernelGGL((ssk::k_features<false, false, true>), grid, block, 0, st, p); break;
        case 5: hipLaunchKernelGGL((ssk::k_features<true, false, true>), grid, block, 0, st, p); break;
        case 6: hipLaunchKernelGGL((ssk::k_features<false, true, true>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((ssk::k_features<true, true, true>), grid, block, 0, st, p); break;
    }
    return hip_err(hipGetLastError());
}

int ss_audio_obs_f32(const float* spec, const float* rir, const int* rir_len, const int* unit_desc,
                     float* audiogoal, float* spectrogram, int n_units, long long rir_unit_stride,
                     int rir_chan_stride, int rir_elem_stride, int rir_cap, int n_valid, int out_len,
                     int pad_mode, int flags, void* stream) {
    if (n_units == 0) return 0;
    if (!spectrogram || n_units < 0) return SS_EINVAL;
    if (pad_mode != SS_PAD_REFLECT && pad_mode != SS_PAD_CONSTANT) return SS_EINVAL;
    if (out_len < ssk::kNfft / 2 + 1) return SS_EINVAL;
    ssk::ConvParams p;
    int n_cus = 1;
    int rc = fill_conv(p, &n_cus, spec, rir, rir_len, unit_desc, rir_unit_stride, rir_chan_stride, rir_elem_stride,
                       rir_cap, n_valid, out_len);
    if (rc) return rc;
    p.pad_mode = pad_mode;
    if (out_len <= ssk::kB && p.t4 <= 26) {             // fused: waveform stays in LDS
        p.out = audiogoal;
        p.sgram = spectrogram;
        return launch_conv<true>(p, n_units, 1, flags, n_cus, static_cast<hipStream_t>(stream));
    }
    if (wide_one_block_ok(out_len, n_valid, flags)) {    // one rendered block of a longer row: fused loop kernel, one launch
        p.out = audiogoal;
        p.sgram = spectrogram;
        return launch_conv<true>(p, n_units, 1, flags, n_cus, static_cast<hipStream_t>(stream), true);
    }
    if (obs_rows_ok(out_len, n_valid, (rir_cap + ssk::kB - 1) / ssk::kB, flags, false, audiogoal != nullptr)) {   // rows of 2-3 blocks: fused as well
        p.out = audiogoal;
        p.sgram = spectrogram;
        return launch_obs_rows<false>(p, n_units, flags, n_cus, static_cast<hipStream_t>(stream));
    }
    if (!audiogoal) return SS_EINVAL;                   // cross-faded long rows hand over through HBM/L2
    rc = ss_fftconv_binaural_f32(spec, rir, rir_len, unit_desc, audiogoal, n_units, rir_unit_stride,
                                 rir_chan_stride, rir_elem_stride, rir_cap, n_valid, out_len, flags, stream);
    if (rc) return rc;
    return spectrogram_of_rows(audiogoal, spectrogram, n_units, out_len, n_valid, pad_mode, stream);
}

// ---- 512-thread FFT core (ss_kernels32.hpp): loop-free rows only; spectra in that core's own register order ----------
int ss_source_windows32_f32(const float* src, const int* win_desc, float* spec_out, int n_windows, void* stream) {
    if (n_windows == 0) return 0;
    if (!src || !win_desc || !spec_out || n_windows < 0) return SS_EINVAL;
    ssk::SrcParams p;
    int rc = get_tables(&p.tb);
    if (rc) return rc;
    p.src = src;
    p.desc = win_desc;
    p.spec = reinterpret_cast<ssk::f32x4*>(spec_out);
    p.desc_stride = 4;
    p.scale = ssk::kWindowScale;
    hipLaunchKernelGGL(ssk::k_source_windows32, dim3(n_windows), dim3(ssk::kT32), 0, static_cast<hipStream_t>(stream), p);
    return hip_err(hipGetLastError());
}

// Rows of ONE block from a time-domain bank of rir_cap <= kB without distractor / cross-fade terms (the loop-free case of
// ss_audio_obs_f32 / ss_fftconv_binaural_f32); spectrogram may be NULL (audiogoal only) or audiogoal may be NULL.
int ss_audio_obs32_f32(const float* spec32, const float* rir, const int* rir_len, const int* unit_desc, float* audiogoal,
                       float* spectrogram, int n_units, long long rir_unit_stride, int rir_chan_stride, int rir_elem_stride,
                       int rir_cap, int n_valid, int out_len, int pad_mode, void* stream) {
    if (n_units == 0) return 0;
    if ((!spectrogram && !audiogoal) || n_units < 0) return SS_EINVAL;
    if (pad_mode != SS_PAD_REFLECT && pad_mode != SS_PAD_CONSTANT) return SS_EINVAL;
    if (n_valid > ssk::kB || rir_cap > ssk::kB || out_len < ssk::kNfft / 2 + 1) return SS_EINVAL;
    ssk::ConvParams p;
    int n_cus = 1;
    int rc = fill_conv(p, &n_cus, spec32, rir, rir_len, unit_desc, rir_unit_stride, rir_chan_stride, rir_elem_stride,
                       rir_cap, n_valid, out_len);
    if (rc) return rc;
    p.pad_mode = pad_mode;
    p.nb_y = 1;
    p.out = audiogoal;
    p.sgram = spectrogram;
    if (spectrogram) {
        if (out_len > ssk::kB || p.t4 > 26) return SS_EINVAL;
        return launch_conv32<true>(p, n_units, static_cast<hipStream_t>(stream));
    }
    return launch_conv32<false>(p, n_units, static_cast<hipStream_t>(stream));
}

int ss_intensity_f32(const float* audiogoal, float* out, int n_units, int len, int num_frame, void* stream) {
    if (n_units == 0) return 0;
    if (!audiogoal || !out || n_units < 0 || len <= 0 || num_frame <= 0) return SS_EINVAL;
    ssk::IntensityParams p;
    p.x = audiogoal;
    p.out = out;
    p.len = len;
    p.num_frame = num_frame;
    hipLaunchKernelGGL(ssk::k_intensity, dim3(n_units), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return hip_err(hipGetLastError());
}


// ---- spectral RIR bank ---------------------------------------------------------------------------------------------
}  // extern "C"
// ss_rir_spectra_f32 for a handful of rows: pinned descriptor blocks (per device), reused behind an event
constexpr int kSmallSpectraWindows = 256;
struct SmallSpectraRing {
    static constexpr int kSlots = 8;
    struct Slot { int* desc = nullptr; hipEvent_t ev = nullptr; } slot[kSlots];
    unsigned next = 0;
};
static std::mutex g_small_mu;
static std::map<int, SmallSpectraRing> g_small_ring;
extern "C" {

int ss_rir_spectra_f32(const float* rir, float* hspec_out, int n_entries, long long rir_unit_stride, int rir_chan_stride,
                       int rir_cap, void* stream) {
    if (n_entries == 0) return 0;
    if (!rir || !hspec_out || n_entries < 0 || rir_cap <= 0 || rir_unit_stride < 0 || rir_chan_stride < 0) return SS_EINVAL;
    const int hb = (rir_cap + ssk::kB - 1) / ssk::kB;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // A HANDFUL of rows (a step's new poses, an eager call's one): descriptors in a small ring of pinned blocks the kernel reads
    // in place - no allocation, no copy, no synchronisation (the general path below does all three per call: ~60 us for one row,
    // which is what an eager call at a new pose paid on top of its file read).  A block is reused after its event has passed.
    if (static_cast<long long>(n_entries) * 2 * hb <= kSmallSpectraWindows &&
        static_cast<long long>(n_entries) * rir_unit_stride < (1LL << 31)) {
        int dev = 0;
        hipError_t e0 = hipGetDevice(&dev);
        if (e0 != hipSuccess) return hip_err(e0);
        std::lock_guard<std::mutex> lk(g_small_mu);
        SmallSpectraRing& ring = g_small_ring[dev];
        SmallSpectraRing::Slot& sl = ring.slot[ring.next++ % SmallSpectraRing::kSlots];
        if (!sl.desc) {
            e0 = hipHostMalloc(reinterpret_cast<void**>(&sl.desc), sizeof(int) * 4 * kSmallSpectraWindows, hipHostMallocDefault);
            if (e0 == hipSuccess) e0 = hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming);
            if (e0 != hipSuccess) { sl.desc = nullptr; (void)hipGetLastError(); }
        } else {
            e0 = hipEventSynchronize(sl.ev);
        }
        if (sl.desc && e0 == hipSuccess) {
            int w = 0;
            for (int r = 0; r < n_entries; ++r)
                for (int c = 0; c < 2; ++c)
                    for (int i = 0; i < hb; ++i, ++w) {
                        const int left = rir_cap - i * ssk::kB;
                        sl.desc[4 * w + 0] = static_cast<int>(r * rir_unit_stride + (long long)c * rir_chan_stride + (long long)i * ssk::kB);
                        sl.desc[4 * w + 1] = left < ssk::kB ? left : ssk::kB;
                        sl.desc[4 * w + 2] = 0;
                        sl.desc[4 * w + 3] = 0;
                    }
            ssk::SrcParams p;
            int rc = get_tables(&p.tb);
            if (rc) return rc;
            p.src = rir;
            p.desc = sl.desc;
            p.spec = reinterpret_cast<ssk::f32x4*>(hspec_out);
            p.desc_stride = 4;
            p.scale = 1.0f;
            hipLaunchKernelGGL(ssk::k_source_windows, dim3(w), dim3(ssk::kT), 0, st, p);
            rc = hip_err(hipGetLastError());
            if (rc == 0) rc = hip_err(hipEventRecord(sl.ev, st));
            return rc;
        }
    }
    // window offsets are int32 words relative to a base pointer: walk the bank in chunks that keep them below 2^31
    long long per = (rir_unit_stride > 0) ? ((1LL << 30) / rir_unit_stride) : n_entries;
    if (per < 1) return SS_EINVAL;
    if (per > 2048) per = 2048;
    std::vector<int> host(static_cast<size_t>(per) * 2 * hb * 4);
    int* dev_desc = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dev_desc), host.size() * sizeof(int));
    if (e != hipSuccess) return hip_err(e);
    int rc = 0;
    for (long long r0 = 0; r0 < n_entries && rc == 0; r0 += per) {
        const int cnt = static_cast<int>(n_entries - r0 < per ? n_entries - r0 : per);
        int w = 0;
        for (int r = 0; r < cnt; ++r)
            for (int c = 0; c < 2; ++c)
                for (int i = 0; i < hb; ++i, ++w) {
                    const int left = rir_cap - i * ssk::kB;
                    host[4 * w + 0] = static_cast<int>(r * rir_unit_stride + (long long)c * rir_chan_stride + (long long)i * ssk::kB);
                    host[4 * w + 1] = left < ssk::kB ? left : ssk::kB;       // the block's samples; zero padding beyond
                    host[4 * w + 2] = 0;
                    host[4 * w + 3] = 0;
                }
        e = hipMemcpyAsync(dev_desc, host.data(), sizeof(int) * 4 * static_cast<size_t>(w), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { rc = hip_err(e); break; }
        ssk::SrcParams p;
        rc = get_tables(&p.tb);
        if (rc) break;
        p.src = rir + r0 * rir_unit_stride;
        p.desc = dev_desc;
        p.spec = reinterpret_cast<ssk::f32x4*>(hspec_out) + static_cast<size_t>(r0) * 2 * hb * (ssk::kSpecComplex / 2);
        p.desc_stride = 4;
        p.scale = 1.0f;                                                   // H' = 2 * rFFT, unscaled (S' carries 1/(8M))
        hipLaunchKernelGGL(ssk::k_source_windows, dim3(w), dim3(ssk::kT), 0, st, p);
        rc = hip_err(hipGetLastError());
        if (rc == 0) rc = hip_err(hipStreamSynchronize(st));             // host[] / dev_desc are reused by the next chunk
    }
    (void)hipFree(dev_desc);
    return rc;
}

}  // extern "C"

template <bool FUSE>
static int launch_conv_spec(ssk::ConvParams p, int n_units, int nb_y, int flags, hipStream_t st, int n_cus = 0) {
    if (nb_y < 1 || nb_y > 3 || (FUSE && nb_y != 1) || (flags & SS_FLAG_CROSSFADE)) return SS_EINVAL;
    p.nb_y = nb_y;
    p.parts_log2 = FUSE && n_cus > 0 ? parts_log2_for(2 * n_units, n_cus) : 0;
    const bool simple = (flags & SS_FLAG_NO_DISTRACTOR) && nb_y == 1 && p.h_blocks == 1 &&
                        (p.n_buckets == 1 || (flags & SS_FLAG_FIRST_BUCKET));
    static const bool no_rows = ab_flag("SS_HIP_NO_ROW_KERNEL");
    if constexpr (!FUSE) {              // more rows than CUs: persistent workgroups prefetching the next row's H'
        if (simple && !no_rows && n_cus > 0 && n_units > n_cus && !(reinterpret_cast<size_t>(p.hspec) & 15)) {
            hipLaunchKernelGGL(ssk::k_conv_spec_rows, dim3(n_cus), dim3(ssk::kT), 0, st, p, 2 * n_units);
            return hip_err(hipGetLastError());
        }
    }
    const dim3 grid((2 * n_units * nb_y) << p.parts_log2), block(ssk::kT);
    if (simple) {
        ssk::UnitTab<true> ut;
        if (fill_unit_tab(ut, g_host_desc, n_units)) hipLaunchKernelGGL((ssk::k_conv_spec<FUSE, true, true>), grid, block, 0, st, p, ut);
        else hipLaunchKernelGGL((ssk::k_conv_spec<FUSE, true>), grid, block, 0, st, p, ssk::UnitTab<false>());
    } else hipLaunchKernelGGL((ssk::k_conv_spec<FUSE, false>), grid, block, 0, st, p, ssk::UnitTab<false>());
    return hip_err(hipGetLastError());
}

extern "C" {

int ss_fftconv_binaural_spec_f32(const float* spec, const float* hspec, const int* rir_len, const int* unit_desc,
                                 float* out, int n_units, int h_blocks, int n_valid, int out_len, int flags,
                                 void* stream) {
    if (n_units == 0) return 0;
    if (!out || !hspec || n_units < 0 || h_blocks < 1) return SS_EINVAL;
    ssk::ConvParams p;
    int n_cus = 1;
    int rc = fill_conv(p, &n_cus, spec, nullptr, rir_len, unit_desc, 0, 0, 1, 0, n_valid, out_len, false);
    if (rc) return rc;
    p.out = out;
    p.hspec = reinterpret_cast<const ssk::f32x4*>(hspec);
    p.h_blocks = h_blocks;
    const int nb_y = n_valid == 0 ? 1 : (n_valid + ssk::kB - 1) / ssk::kB;
    return launch_conv_spec<false>(p, n_units, nb_y, flags, static_cast<hipStream_t>(stream), n_cus);
}

int ss_audio_obs_spec_f32(const float* spec, const float* hspec, const int* rir_len, const int* unit_desc,
                          float* audiogoal, float* spectrogram, int n_units, int h_blocks, int n_valid, int out_len,
                          int pad_mode, int flags, void* stream) {
    if (n_units == 0) return 0;
    if (!spectrogram || !hspec || n_units < 0 || h_blocks < 1) return SS_EINVAL;
    if (pad_mode != SS_PAD_REFLECT && pad_mode != SS_PAD_CONSTANT) return SS_EINVAL;
    if (out_len < ssk::kNfft / 2 + 1) return SS_EINVAL;
    ssk::ConvParams p;
    int n_cus = 1;
    int rc = fill_conv(p, &n_cus, spec, nullptr, rir_len, unit_desc, 0, 0, 1, 0, n_valid, out_len, false);
    if (rc) return rc;
    p.pad_mode = pad_mode;
    p.hspec = reinterpret_cast<const ssk::f32x4*>(hspec);
    p.h_blocks = h_blocks;
    if (out_len <= ssk::kB && p.t4 <= 26) {
        p.out = audiogoal;
        p.sgram = spectrogram;
        return launch_conv_spec<true>(p, n_units, 1, flags, static_cast<hipStream_t>(stream), n_cus);
    }
    if (obs_rows_ok(out_len, n_valid, h_blocks, flags, true, audiogoal != nullptr)) {
        p.out = audiogoal;
        p.sgram = spectrogram;
        return launch_obs_rows<true>(p, n_units, flags, n_cus, static_cast<hipStream_t>(stream));
    }
    if (!audiogoal) return SS_EINVAL;
    rc = ss_fftconv_binaural_spec_f32(spec, hspec, rir_len, unit_desc, audiogoal, n_units, h_blocks, n_valid, out_len,
                                      flags, stream);
    if (rc) return rc;
    return spectrogram_of_rows(audiogoal, spectrogram, n_units, out_len, n_valid, pad_mode, stream);
}

// ---- context API (include/ss_hip.h): planner + window-spectra cache + descriptor ring inside the library --------------
struct ss_ctx { ssctx::Context c; };

int ss_ctx_create(ss_ctx** out, int sampling_rate, int n_valid, int pad_mode, int wrap_mode, int max_window_sets) {
    if (!out || sampling_rate <= 0 || n_valid < 0 || n_valid > sampling_rate || n_valid > 3 * ssk::kB) return SS_EINVAL;
    if (pad_mode != SS_PAD_REFLECT && pad_mode != SS_PAD_CONSTANT) return SS_EINVAL;
    ss_ctx* h = new (std::nothrow) ss_ctx();
    if (!h) return SS_EINVAL;
    ssctx::Context& c = h->c;
    c.sr = sampling_rate; c.n_valid = n_valid; c.out_len = sampling_rate; c.pad_mode = pad_mode;
    c.wrap_mode = wrap_mode ? 1 : 0;
    c.kb = ssk::kB; c.spec_floats = 2 * ssk::kSpecComplex;
    c.n_entries = max_window_sets > 0 ? max_window_sets : 256;
    c.stride = n_valid > 0 ? ssctx::ceil_div(n_valid, ssk::kB) : 1;                 // nbh_max (= 1 until a bank is set) + nby - 1
    ssctx::cache_reset(c);
    *out = h;
    return 0;
}

static void ctx_free_device(ssctx::Context& c) {
    if (c.pool) (void)hipFree(c.pool);
    if (c.src_dev) (void)hipFree(c.src_dev);
    if (c.d_desc) (void)hipFree(c.d_desc);
    if (c.d_win) (void)hipFree(c.d_win);
    if (c.h_desc) (void)hipHostFree(c.h_desc);
    if (c.h_win) (void)hipHostFree(c.h_win);
    if (c.ag_scratch) (void)hipFree(c.ag_scratch);
    c.ag_scratch = nullptr; c.ag_cap = 0;
    if (c.ev_made)
        for (int k = 0; k < ssctx::kRing / ssctx::kGroup; ++k) (void)hipEventDestroy(c.ev_done[k]);
    if (c.ev_xstream) (void)hipEventDestroy(c.ev_xstream);
    if (c.miss_ev) { (void)hipEventDestroy(static_cast<hipEvent_t>(c.miss_ev)); c.miss_ev = nullptr; }
    c.ev_xstream = nullptr; c.have_last_stream = false;
    if (c.lanes_made) {
        for (int l = 0; l < ssctx::kLanes; ++l) {
            (void)hipStreamSynchronize(c.lane_stream[l]);
            drop_stash(c.lane_stream[l]);                       // k_obs_rows' per-(device, stream) scratch of this lane
            (void)hipEventDestroy(c.ev_lane[l]);
            (void)hipEventDestroy(c.ev_win[l]);
            (void)hipStreamDestroy(c.lane_stream[l]);
        }
        (void)hipEventDestroy(c.ev_in);
        for (int l = 1; l < ssctx::kLanes; ++l)
            for (int k = 0; k < ssctx::kRing / ssctx::kGroup; ++k) (void)hipEventDestroy(c.ev_done_l[l][k]);
        c.lanes_made = false;
    }
    c.pool = nullptr; c.src_dev = nullptr; c.d_desc = nullptr; c.d_win = nullptr; c.h_desc = nullptr; c.h_win = nullptr;
    c.ev_made = false;
}

int ss_ctx_destroy(ss_ctx* h) {
    if (!h) return 0;
    ctx_free_device(h->c);
    delete h;
    return 0;
}

// `clip` is HOST memory unless on_device != 0.  Returns the sound id (>= 0) or a negative error.
int ss_ctx_add_source(ss_ctx* h, const float* clip, int len, int on_device) {
    if (!h || !clip || len <= 0) return SS_EINVAL;
    ssctx::Context& c = h->c;
    const size_t need = c.src_used + static_cast<size_t>(len);
    if (need > c.src_cap) {                                    // grow the flat bank (rare: once per new sound at most)
        size_t cap = c.src_cap ? c.src_cap : (1u << 20);
        while (cap < need) cap *= 2;
        float* nb = nullptr;
        hipError_t e = hipMalloc(&nb, cap * sizeof(float));
        if (e != hipSuccess) return hip_err(e);
        if (c.src_used) {
            e = hipDeviceSynchronize();                        // launches that read the old bank
            if (e == hipSuccess) e = hipMemcpy(nb, c.src_dev, c.src_used * sizeof(float), hipMemcpyDeviceToDevice);
            if (e != hipSuccess) { (void)hipFree(nb); return hip_err(e); }
        }
        if (c.src_dev) (void)hipFree(c.src_dev);
        c.src_dev = nb;
        c.src_cap = cap;
    }
    hipError_t e = hipMemcpy(c.src_dev + c.src_used, clip, static_cast<size_t>(len) * sizeof(float),
                             on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_err(e);
    c.src_off.push_back(static_cast<int>(c.src_used));
    c.src_len.push_back(len);
    c.src_used = need;
    return static_cast<int>(c.src_len.size()) - 1;
}

// planning-only registration (tests of the planner on machines without a GPU): lengths only, no device memory
int ss_ctx_add_source_len(ss_ctx* h, int len) {
    if (!h || len <= 0) return SS_EINVAL;
    ssctx::Context& c = h->c;
    c.src_off.push_back(static_cast<int>(c.src_used));
    c.src_len.push_back(len);
    c.src_used += static_cast<size_t>(len);
    return static_cast<int>(c.src_len.size()) - 1;
}

int ss_ctx_set_rir_bank(ss_ctx* h, const float* rir, const int* rir_len, long long unit_stride, int chan_stride,
                        int elem_stride, int rir_cap) {
    if (!h || rir_cap < 0 || elem_stride < 1 || chan_stride < 0 || unit_stride < 0) return SS_EINVAL;
    ssctx::Context& c = h->c;
    const int nbh_old = c.rir_cap > 0 ? ssctx::ceil_div(c.rir_cap, c.kb) : 1;
    const int nbh_new = rir_cap > 0 ? ssctx::ceil_div(rir_cap, c.kb) : 1;
    c.rir = rir; c.rir_len = rir_len; c.rir_us = unit_stride; c.rir_cs = chan_stride; c.rir_es = elem_stride;
    c.rir_cap = rir_cap;
    // the spectral form described the PREVIOUS bank (its entries, its capacity): back to the time-domain kernels until
    // ss_ctx_set_rir_spectra is called for this one (a swapped / grown bank kept rendering the old spectra)
    c.hspec = nullptr;
    c.h_blocks = 0;
    c.buckets.clear();
    if (nbh_new != nbh_old) {                                  // the set of partition offsets per key changes
        const int nby = c.n_valid > 0 ? ssctx::ceil_div(c.n_valid, c.kb) : 1;
        c.stride = nbh_new + nby - 1;
        ssctx::cache_reset(c);
        if (c.pool) {
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) return hip_err(e);
            (void)hipFree(c.pool);
            c.pool = nullptr;
            c.pool_entries = 0;
        }
    }
    return 0;
}

// Spectral form of the bank set by ss_ctx_set_rir_bank (same entries, same rir_len): steps without a cross-fade then
// run k_conv_spec.  hspec = NULL switches back to the time-domain kernels.
int ss_ctx_set_rir_spectra(ss_ctx* h, const float* hspec, int h_blocks) {
    if (!h || (hspec && h_blocks < 1)) return SS_EINVAL;
    ssctx::Context& c = h->c;
    if (hspec && h_blocks != (c.rir_cap > 0 ? ssctx::ceil_div(c.rir_cap, c.kb) : 1)) return SS_EINVAL;
    c.hspec = hspec;
    c.h_blocks = hspec ? h_blocks : 0;
    return 0;
}

int ss_ctx_plan(ss_ctx* h, const ss_units* units, int n, int* unit_desc_out, int* flags_out, int* n_new_windows_out,
                int* new_windows_out, int new_windows_cap) {
    if (!h || !unit_desc_out) return SS_EINVAL;
    ssctx::PlanResult res;
    int rc = ssctx::plan_units(h->c, units, n, unit_desc_out, &res);
    if (rc) return rc;
    h->c.plan_only_keys = true;       // keys in the cache whose spectra nobody computes, a tick without a ring slot: the
                                      // next ss_ctx_observe starts from an empty cache (after a device synchronise)
    if (flags_out) *flags_out = res.flags;
    if (n_new_windows_out) *n_new_windows_out = res.n_new_windows;
    if (new_windows_out) {
        const int k = res.n_new_windows < new_windows_cap ? res.n_new_windows : new_windows_cap;
        std::memcpy(new_windows_out, h->c.new_win.data(), sizeof(int) * 5 * static_cast<size_t>(k));
    }
    return 0;
}

int ss_ctx_stats(ss_ctx* h, long long* out8) {
    if (!h || !out8) return SS_EINVAL;
    const ssctx::Context& c = h->c;
    out8[0] = c.hits; out8[1] = c.misses; out8[2] = c.evictions; out8[3] = c.grows;
    out8[4] = c.n_entries; out8[5] = static_cast<long long>(c.map.size()); out8[6] = c.stride; out8[7] = c.tick;
    return 0;
}

static int ctx_ensure_ring(ssctx::Context& c, int n, int n_win, hipStream_t st) {
    hipError_t e;
    if (!c.ev_made) {
        for (int k = 0; k < ssctx::kRing / ssctx::kGroup; ++k) {
            e = hipEventCreateWithFlags(&c.ev_done[k], hipEventDisableTiming);
            if (e != hipSuccess) return hip_err(e);
        }
        c.ev_made = true;
    }
    if (n > c.ring_cap) {
        e = hipDeviceSynchronize();                            // slots of the old ring may still be read
        if (e != hipSuccess) return hip_err(e);
        int cap = c.ring_cap ? c.ring_cap : 256;
        while (cap < n) cap *= 2;
        if (c.h_desc) (void)hipHostFree(c.h_desc);
        if (c.d_desc) (void)hipFree(c.d_desc);
        c.h_desc = nullptr; c.d_desc = nullptr; c.ring_cap = 0;
        const size_t bytes = sizeof(int) * 8 * static_cast<size_t>(cap) * ssctx::kRing;
        e = hipHostMalloc(reinterpret_cast<void**>(&c.h_desc), bytes, hipHostMallocDefault);
        if (e != hipSuccess) return hip_err(e);
        e = hipMalloc(reinterpret_cast<void**>(&c.d_desc), bytes);
        if (e != hipSuccess) return hip_err(e);
        c.ring_cap = cap;
    }
    if (n_win > c.win_cap) {
        e = hipDeviceSynchronize();
        if (e != hipSuccess) return hip_err(e);
        int cap = c.win_cap ? c.win_cap : 256;
        while (cap < n_win) cap *= 2;
        if (c.h_win) (void)hipHostFree(c.h_win);
        if (c.d_win) (void)hipFree(c.d_win);
        c.h_win = nullptr; c.d_win = nullptr; c.win_cap = 0;
        const size_t bytes = sizeof(int) * 5 * static_cast<size_t>(cap) * ssctx::kRing;
        e = hipHostMalloc(reinterpret_cast<void**>(&c.h_win), bytes, hipHostMallocDefault);
        if (e != hipSuccess) return hip_err(e);
        e = hipMalloc(reinterpret_cast<void**>(&c.d_win), bytes);
        if (e != hipSuccess) return hip_err(e);
        c.win_cap = cap;
    }
    (void)st;
    return 0;
}

static int ctx_ensure_pool(ssctx::Context& c, hipStream_t st) {
    if (c.pool && c.pool_entries >= c.n_entries) return 0;
    const size_t per_entry = static_cast<size_t>(c.stride) * c.spec_floats * sizeof(float);
    float* np = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&np), per_entry * static_cast<size_t>(c.n_entries));
    if (e != hipSuccess) return hip_err(e);
    if (c.pool) {                                              // grown: keep the spectra already computed
        e = hipMemcpyAsync(np, c.pool, per_entry * static_cast<size_t>(c.pool_entries), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipDeviceSynchronize();       // earlier launches (any stream) still read the old pool
        if (e != hipSuccess) { (void)hipFree(np); return hip_err(e); }
        (void)hipFree(c.pool);
    }
    c.pool = np;
    c.pool_entries = c.n_entries;
    return 0;
}

// One step: plan the units (host), compute the missing source-window spectra, render.  `units` are HOST arrays.
// audiogoal / spectrogram are DEVICE buffers [n,2,sr] / [n,65,T4,2]; either may be NULL (not both).
// `lane` >= 0: the step runs on internal stream `lane` of the overlap mode (stream == c.lane_stream[lane])
static int ctx_observe_on(ss_ctx* h, const ss_units* units, int n, float* audiogoal, float* spectrogram, void* stream,
                          int lane) {
    ssctx::Context& c = h->c;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e;
    SS_PROF_BEGIN();
    if (c.plan_only_keys) {                                    // ss_ctx_plan was used on this context: its keys claim
        // spectra that were never computed, and its ticks took no ring slot (the eviction guard of the overlap mode counts
        // ticks as ring slots) - start from an empty cache, and only once nothing in flight reads the pool (a
        // reset under two lanes let lane A's k_source_windows write slots lane B was still reading)
        e = hipDeviceSynchronize();
        if (e != hipSuccess) return hip_err(e);
        ssctx::cache_reset(c);
        c.plan_only_keys = false;
    }
    // The window-spectra pool is shared by every step: a step on a NEW stream reads spectra the previous stream wrote
    // (cache hits) and may overwrite slots it still reads (evictions).  One event orders the new stream behind everything
    // the old one has been given so far.
    if (lane >= 0) {
        // overlap mode: the lanes only need each other's WINDOW SPECTRA (a cache hit on lane B of a key lane A computed);
        // evictions cannot touch a slot a step in flight reads (guard of a full ring of ticks, ss_context.hpp)
        for (int o = 0; o < c.n_lanes; ++o) {
            if (o == lane || c.win_seen[lane][o] >= c.win_seq[o]) continue;
            e = hipStreamWaitEvent(st, c.ev_win[o], 0);
            if (e != hipSuccess) return hip_err(e);
            c.win_seen[lane][o] = c.win_seq[o];
        }
    } else if (c.have_last_stream && st != c.last_stream) {
        if (!c.ev_xstream) {
            e = hipEventCreateWithFlags(&c.ev_xstream, hipEventDisableTiming);
            if (e != hipSuccess) return hip_err(e);
        }
        e = hipEventRecord(c.ev_xstream, c.last_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, c.ev_xstream, 0);
        if (e != hipSuccess) return hip_err(e);
    }
    SS_PROF_MARK(1);                                           // cross-lane / cross-stream waits
    int rc = ctx_ensure_ring(c, n, 0, st);
    if (rc) return rc;
    // Ring slots are released in GROUPS: one completion event per kGroup consecutive steps (recorded after the group's
    // last launch, waited for - on the host - before the group's first slot is written again, a full ring later).  An
    // event record per step puts a marker packet between every two launches (measured: ~2 us of a 26-us step).  A caller
    // that changes streams inside a group closes it on the old stream and starts the next group.
    if (lane < 0 && c.group_open && (st != c.group_stream || c.ring_k / ssctx::kGroup != c.open_group)) {
        e = hipEventRecord(c.ev_done[c.open_group], c.group_stream);     // (also a group a failed step left open)
        if (e != hipSuccess) return hip_err(e);
        c.group_open = false;
        c.ring_k = (c.open_group + 1) * ssctx::kGroup % ssctx::kRing;
    }
    const int k = c.ring_k, g = k / ssctx::kGroup;
    if (k % ssctx::kGroup == 0) {
        e = hipEventSynchronize(c.ev_done[g]);                 // every launch that read the group's slots has finished
        for (int l = 1; lane >= 0 && l < c.n_lanes && e == hipSuccess; ++l) e = hipEventSynchronize(c.ev_done_l[l][g]);   // ... on every lane
        if (e != hipSuccess) return hip_err(e);
        c.group_open = lane < 0;
        c.group_stream = st;
        c.open_group = g;
    }
    // Small steps: the kernels read the unit descriptors straight from the pinned ring slot (one 32-byte scalar load per
    // workgroup over the host link) - an upload between two launches on the stream costs a blit kernel plus a barrier on
    // either side of it (measured: ~15 us of idle GPU per 25-us step).  Large steps (several descriptors per workgroup,
    // long kernels) keep the upload.
    static const bool force_copy = ab_flag("SS_HIP_DESC_COPY");
    const bool direct = !force_copy && n <= ssctx::kDirectDescUnits;
    int* hd = c.h_desc + static_cast<size_t>(k) * c.ring_cap * 8;
    int* dd = direct ? hd : c.d_desc + static_cast<size_t>(k) * c.ring_cap * 8;
    ssctx::PlanResult res;
    SS_PROF_MARK(2);                                           // ring slot: group event synchronise
    rc = ssctx::plan_units(c, units, n, hd, &res);
    if (rc) return rc;                                         // (refused before the cache or the ring was touched)
    SS_PROF_MARK(3);                                           // planner
    c.ring_k = (k + 1) % ssctx::kRing;                         // the slot is taken from here on, whatever happens next
    // The slot's group is released by an event recorded behind the group's last launch (overlap mode: behind each lane's
    // last launch of the group); a step that fails after this point must still record it, or the group's next round
    // would be checked against a stale event.
    auto close_slot = [&]() -> int {
        hipError_t ee = hipSuccess;
        if (lane >= 0) {
            if (k % ssctx::kGroup >= ssctx::kGroup - c.n_lanes) ee = hipEventRecord(lane == 0 ? c.ev_done[g] : c.ev_done_l[lane][g], st);
        } else if (k % ssctx::kGroup == ssctx::kGroup - 1) {
            ee = hipEventRecord(c.ev_done[g], st);
            c.group_open = false;
        }
        return hip_err(ee);
    };
    // ... and take the keys of this plan out of the cache again as long as their spectra have not been computed
    auto fail = [&](int code) { ssctx::cache_rollback(c); (void)close_slot(); return code; };
    rc = ctx_ensure_pool(c, st);
    if (rc) return fail(rc);
    if (res.n_new_windows > 0) {
        rc = ctx_ensure_ring(c, n, res.n_new_windows, st);
        if (rc) return fail(rc);
        int* hw = c.h_win + static_cast<size_t>(k) * c.win_cap * 5;
        int* dw = c.d_win + static_cast<size_t>(k) * c.win_cap * 5;
        std::memcpy(hw, c.new_win.data(), sizeof(int) * 5 * static_cast<size_t>(res.n_new_windows));
        e = hipMemcpyAsync(dw, hw, sizeof(int) * 5 * static_cast<size_t>(res.n_new_windows), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return fail(hip_err(e));
        rc = launch_windows_scatter(c.src_dev, dw, c.pool, res.n_new_windows, st);
        if (rc) return fail(rc);
        if (lane >= 0) {                                       // the other lane's next step may hit these keys
            e = hipEventRecord(c.ev_win[lane], st);
            if (e != hipSuccess) return fail(hip_err(e));
            ++c.win_seq[lane];
        }
    }
    c.new_entries.clear();                                     // their spectra are on the stream: the keys are good
    c.last_stream = st;
    c.have_last_stream = true;
    if (!direct) {
        e = hipMemcpyAsync(dd, hd, sizeof(int) * 8 * static_cast<size_t>(n), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return fail(hip_err(e));
    }
    // cross-faded steps take the time-domain rows; so do LARGE steps of one-block rows when the caller keeps both forms and said
    // so (ss_ctx_set_spectral_policy): there the forward FFT hides under the row's load and the spectral rows are twice the bytes
    // (... unless its units carry a second term - a distractor, simulator.py:649-664: two forward transforms per row do not hide
    // under one row's load; savi's 256-env step: 87.9 against 96.3 us)
    const bool spectral = c.hspec && !(res.flags & SS_FLAG_CROSSFADE) &&
                          !(c.spectral_max_units > 0 && c.rir && c.out_len <= ssk::kB && n > c.spectral_max_units &&
                            (res.flags & SS_FLAG_NO_DISTRACTOR));
    const int nbh_bank = spectral ? c.h_blocks : (c.rir_cap > 0 ? ssctx::ceil_div(c.rir_cap, c.kb) : 1);
    if (spectrogram && !audiogoal && c.out_len > ssk::kB && !wide_one_block_ok(c.out_len, c.n_valid, res.flags, spectral) &&
        !obs_rows_ok(c.out_len, c.n_valid, nbh_bank, res.flags, spectral, true)) {  // cross-faded / very long rows hand over through memory (the context's own buffer)
        const size_t need = static_cast<size_t>(n) * 2 * c.out_len;
        if (need > c.ag_cap) {
            e = hipDeviceSynchronize();
            if (e != hipSuccess) return fail(hip_err(e));
            if (c.ag_scratch) (void)hipFree(c.ag_scratch);
            c.ag_scratch = nullptr; c.ag_cap = 0;
            e = hipMalloc(reinterpret_cast<void**>(&c.ag_scratch), need * sizeof(float));
            if (e != hipSuccess) return fail(hip_err(e));
            c.ag_cap = need;
        }
        audiogoal = c.ag_scratch;
    }
    static const bool no_tab = ab_flag("SS_HIP_NO_UNIT_TAB");
    SS_PROF_MARK(4);                                           // new windows (upload + k_source_windows), descriptor upload
    g_host_desc = no_tab ? nullptr : hd;                       // (see fill_unit_tab; cleared right after the dispatch below)
    g_launch_share = c.chip_share > 0 ? c.chip_share : c.n_lanes;
    if (!c.buckets.empty()) {
        const int nb = static_cast<int>(c.buckets.size());
        rc = spectrogram ? ss_audio_obs_buckets_f32(c.pool, c.buckets.data(), nb, c.rir_len, dd, audiogoal, spectrogram, n,
                                                    c.n_valid, c.out_len, c.pad_mode, res.flags, stream)
                         : ss_fftconv_binaural_buckets_f32(c.pool, c.buckets.data(), nb, c.rir_len, dd, audiogoal, n, c.n_valid,
                                                           c.out_len, res.flags, stream);
    } else if (spectrogram && spectral)
        rc = ss_audio_obs_spec_f32(c.pool, c.hspec, c.rir_len, dd, audiogoal, spectrogram, n, c.h_blocks, c.n_valid,
                                   c.out_len, c.pad_mode, res.flags, stream);
    else if (spectrogram)
        rc = ss_audio_obs_f32(c.pool, c.rir, c.rir_len, dd, audiogoal, spectrogram, n, c.rir_us, c.rir_cs, c.rir_es,
                              c.rir_cap, c.n_valid, c.out_len, c.pad_mode, res.flags, stream);
    else if (spectral)
        rc = ss_fftconv_binaural_spec_f32(c.pool, c.hspec, c.rir_len, dd, audiogoal, n, c.h_blocks, c.n_valid, c.out_len,
                                          res.flags, stream);
    else
        rc = ss_fftconv_binaural_f32(c.pool, c.rir, c.rir_len, dd, audiogoal, n, c.rir_us, c.rir_cs, c.rir_es, c.rir_cap,
                                     c.n_valid, c.out_len, res.flags, stream);
    g_host_desc = nullptr;
    g_launch_share = 1;
    SS_PROF_MARK(5);                                           // the launch entry (unit table + hipLaunchKernel)
    if (rc) return fail(rc);
    // (overlap mode: a group's ticks alternate between the lanes; each lane records its half after ITS last tick)
    rc = close_slot();
    SS_PROF_MARK(6);                                           // group event record
    return rc;
}

// ---- overlap mode -------------------------------------------------------------------------------------------------
int ss_ctx_set_overlap(ss_ctx* h, int n_streams) {
    if (!h || n_streams < 1 || n_streams > ssctx::kLanes) return SS_EINVAL;
    ssctx::Context& c = h->c;
    hipError_t e = hipDeviceSynchronize();                     // a clean cut between the two regimes
    if (e != hipSuccess) return hip_err(e);
    if (n_streams > 1 && !c.lanes_made) {
        for (int l = 0; l < ssctx::kLanes; ++l) {
            e = hipStreamCreateWithFlags(&c.lane_stream[l], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ev_lane[l], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ev_win[l], hipEventDisableTiming);
            if (e != hipSuccess) return hip_err(e);
        }
        e = hipEventCreateWithFlags(&c.ev_in, hipEventDisableTiming);
        if (e != hipSuccess) return hip_err(e);
        for (int k = 0; k < ssctx::kRing / ssctx::kGroup; ++k) {
            for (int l = 1; l < ssctx::kLanes; ++l) {
                e = hipEventCreateWithFlags(&c.ev_done_l[l][k], hipEventDisableTiming);
                if (e != hipSuccess) return hip_err(e);
            }
        }
        c.lanes_made = true;
    }
    c.n_lanes = n_streams;
    c.lane_next = 0;
    c.ring_k = 0;                                              // groups start afresh (everything has completed)
    c.group_open = false;
    c.have_last_stream = false;
    for (int l = 0; l < ssctx::kLanes; ++l) { c.lane_dirty[l] = false; c.lane_joined[l] = false; c.lane_join_valid[l] = false; for (int o = 0; o < ssctx::kLanes; ++o) c.win_seen[l][o] = c.win_seq[o]; }
    return 0;
}

int ss_ctx_set_spectral_policy(ss_ctx* h, int max_units) {
    if (!h || max_units < 0) return SS_EINVAL;
    h->c.spectral_max_units = max_units;
    return 0;
}

int ss_ctx_set_chip_share(ss_ctx* h, int n_sources) {
    if (!h || n_sources < 0 || n_sources > 8) return SS_EINVAL;
    h->c.chip_share = n_sources;
    return 0;
}

int ss_ctx_join(ss_ctx* h, void* stream) {
    if (!h) return SS_EINVAL;
    ssctx::Context& c = h->c;
    if (c.n_lanes <= 1) return 0;                              // single-stream mode: the caller's stream IS the work's stream
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int l = 0; l < c.n_lanes; ++l) {
        // a lane with nothing issued since its last join: its event still stands for "everything on the lane" - a join from
        // ANOTHER stream waits for that record, the stream that joined last has nothing left to wait for
        if (!c.lane_dirty[l] && (!c.lane_joined[l] || (c.lane_join_valid[l] && c.lane_join_stream[l] == st))) continue;
        hipError_t e = c.lane_dirty[l] ? hipEventRecord(c.ev_lane[l], c.lane_stream[l]) : hipSuccess;
        if (e == hipSuccess) e = hipStreamWaitEvent(st, c.ev_lane[l], 0);
        if (e != hipSuccess) return hip_err(e);
        c.lane_dirty[l] = false;
        c.lane_joined[l] = true;
        c.lane_join_valid[l] = true;
        c.lane_join_stream[l] = st;
    }
    return 0;
}

// Overlap mode: work on `st` that REWRITES bank rows (the in-call loaders' scatter) goes behind every step issued so far - steps
// in flight on the lanes may still read the entries (write after read).  The lanes' join state is left alone.
static int order_behind_lanes(ssctx::Context& c, hipStream_t st) {
    if (c.n_lanes <= 1) return 0;
    for (int l = 0; l < c.n_lanes; ++l) {
        if (!c.lane_dirty[l] && !c.lane_joined[l]) continue;
        hipError_t e = c.lane_dirty[l] ? hipEventRecord(c.ev_lane[l], c.lane_stream[l]) : hipSuccess;
        if (e == hipSuccess) e = hipStreamWaitEvent(st, c.ev_lane[l], 0);
        if (e != hipSuccess) return hip_err(e);
        if (c.lane_dirty[l]) {                                 // the record now covers the lane's work: a later join re-uses it
            c.lane_dirty[l] = false;
            c.lane_joined[l] = true;
            c.lane_join_valid[l] = false;                      // (no stream has JOINED: every join still waits)
        }
    }
    return 0;
}

static int ctx_features_on(ss_ctx* h, int n, const float* audiogoal, float* spectrogram, const ss_features* f, void* stream) {
    if (!f) return 0;
    const ssctx::Context& c = h->c;
    return ss_audio_features_f32(audiogoal, n, c.out_len, c.pad_mode, spectrogram, f->logmel, f->mel_start, f->mel_w, f->n_mels,
                                 f->max_len, f->mel_eps, f->gccphat, f->max_lag, f->gcc_eps, stream);
}

// A step with extension features: k_features has every frame's spectrum of both ears in registers anyway, so the pooled
// spectrogram comes from IT (+4.7 us per 256 units) and the convolution launch drops its fused STFT phase (-7 us per round of
// 256 rows): the waveform is the hand-over either way (the features need it in memory).
static bool features_take_spectrogram(const ss_features* f, const float* audiogoal, const float* spectrogram) {
    static const bool off = ab_flag("SS_HIP_FEAT_KEEP_FUSED");       // (A/B builds only)
    return !off && f && audiogoal && spectrogram;
}

static int ctx_observe_any(ss_ctx* h, const ss_units* units, int n, float* audiogoal, float* spectrogram, const ss_features* f,
                           void* stream) {
    if (!h || n < 0 || (!audiogoal && !spectrogram)) return SS_EINVAL;
    if (f && (!audiogoal || (!f->logmel && !f->gccphat))) return SS_EINVAL;
    if (n == 0) return 0;
    ssctx::Context& c = h->c;
    if ((!c.rir && !c.hspec && c.buckets.empty()) || !c.rir_len || !c.src_dev) return SS_EINVAL;
    const bool sg_late = features_take_spectrogram(f, audiogoal, spectrogram);
    if (c.n_lanes <= 1) {
        const int rc = ctx_observe_on(h, units, n, audiogoal, sg_late ? nullptr : spectrogram, stream, -1);
        return rc ? rc : ctx_features_on(h, n, audiogoal, sg_late ? spectrogram : nullptr, f, stream);
    }
    // overlap mode: this step goes to the next internal stream, behind whatever the caller's stream holds right now (the
    // consumers of the output rows it overwrites, uploads of RIR rows it reads); the caller's stream sees the result after
    // ss_ctx_join.  Consecutive steps run on different streams: the head of step k+1 (descriptor + row loads, HBM latency,
    // nothing to compute) overlaps the tail of step k (STFT, no memory traffic).
    const int lane = c.ring_k % c.n_lanes;                     // tick parity: a refused call does not shift the lanes
    SS_PROF_BEGIN();
    // The fence costs an event record + a stream wait (measured: 4.6 us of the call's 10.8 us of host time at 16-32 envs, where
    // the host IS the step's bound) and a barrier packet in front of the lane's launch.  A caller's stream that has nothing
    // pending - a trainer that has just read its actions back, a vector env between two policy steps - needs none.
    static const bool always_fence = ab_flag("SS_HIP_ALWAYS_FENCE");        // (A/B builds only)
    //  a stream being captured into a graph must not be QUERIED - that is an illegal operation which invalidates
    // the capture: ask the capture status first and go straight to the fence in that case.  Single-threaded use of `stream` is
    // assumed between the query and the lane's launch (include/ss_hip.h, ss_ctx_set_overlap).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (!always_fence && hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusActive;                    // cannot tell: do not query, fence
    }
    hipError_t e = (always_fence || cap != hipStreamCaptureStatusNone) ? hipErrorNotReady
                                                                       : hipStreamQuery(static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        // work pending (hipErrorNotReady: the only answer that is cleared here - a genuine sticky error stays for the fence's
        // own calls and the launch check to report): order the lane behind the caller's stream
        if (!always_fence && cap == hipStreamCaptureStatusNone && e == hipErrorNotReady) (void)hipGetLastError();
        e = hipEventRecord(c.ev_in, static_cast<hipStream_t>(stream));
        if (e == hipSuccess) e = hipStreamWaitEvent(c.lane_stream[lane], c.ev_in, 0);
    }
    if (e != hipSuccess) return hip_err(e);
    SS_PROF_MARK(0);                                           // input fence: caller's stream -> lane
    SS_PROF_CALL();
    c.lane_dirty[lane] = true;
    const int rc = ctx_observe_on(h, units, n, audiogoal, sg_late ? nullptr : spectrogram, c.lane_stream[lane], lane);
    return rc ? rc : ctx_features_on(h, n, audiogoal, sg_late ? spectrogram : nullptr, f, c.lane_stream[lane]);
}

#if defined(SS_AB)
int ss_ab_host_profile(double* ns_out8, long long* calls_out) {
    for (int i = 0; i < 8; ++i) { ns_out8[i] = g_prof_ns[i]; g_prof_ns[i] = 0; }
    *calls_out = g_prof_calls; g_prof_calls = 0;
    return 0;
}
#endif

int ss_ctx_observe(ss_ctx* h, const ss_units* units, int n, float* audiogoal, float* spectrogram, void* stream) {
    return ctx_observe_any(h, units, n, audiogoal, spectrogram, nullptr, stream);
}

int ss_ctx_observe_features(ss_ctx* h, const ss_units* units, int n, float* audiogoal, float* spectrogram, const ss_features* f,
                            void* stream) {
    if (!f) return SS_EINVAL;
    return ctx_observe_any(h, units, n, audiogoal, spectrogram, f, stream);
}


}  // extern "C"
static int sims_to_units(ss_ctx* h, const ss_sim_columns* sc, int n, int* w, int* miss_out, int* n_miss);
extern "C" {

// One step from the simulators' state columns (see include/ss_hip.h): the per-step numpy of ss_amd/vector.py in C++.
int ss_ctx_sims_units(ss_ctx* h, const ss_sim_columns* sc, int n, int* units_out, int* miss_out, int* n_miss) {
    if (!units_out) return SS_EINVAL;
    return sims_to_units(h, sc, n, units_out, miss_out, n_miss);
}

int ss_ctx_observe_sims(ss_ctx* h, const ss_sim_columns* sc, int n, float* audiogoal, float* spectrogram, int* miss_out,
                        int* n_miss, void* stream) {
    if (!h || n < 0) return SS_EINVAL;
    std::vector<int>& w = h->c.sim_scratch;
    w.resize(static_cast<size_t>(n) * 5 + 1);
    const int rc = sims_to_units(h, sc, n, w.data(), miss_out, n_miss);
    if (rc != 0 || n == 0 || *n_miss) return rc;
    ss_units u;
    std::memset(&u, 0, sizeof u);
    u.sound = w.data(); u.t0 = u.sound + n; u.rir = u.t0 + n;
    if (sc->dis_sound) { u.dis_sound = u.rir + n; u.dis_rir = u.dis_sound + n; }
    return ss_ctx_observe(h, &u, n, audiogoal, spectrogram, stream);
}

}  // extern "C"

static int sims_to_units(ss_ctx* h, const ss_sim_columns* sc, int n, int* w, int* miss_out, int* n_miss) {
    if (!h || !sc || n < 0 || !n_miss || !sc->sound || !sc->audio_index || !sc->step_count || !sc->duration || !sc->recv ||
        !sc->src || !sc->rot || !sc->scene || !sc->index_flat || !sc->index_off || !sc->index_dim || sc->azimuths < 1 ||
        360 % sc->azimuths || (!sc->dis_sound != !sc->dis_src))
        return SS_EINVAL;
    ssctx::Context& c = h->c;
    *n_miss = 0;
    if (n == 0) return 0;
    const bool dis = sc->dis_sound != nullptr;
    const int n_src = static_cast<int>(c.src_len.size()), step = 360 / sc->azimuths;
    int* sound = w; int* t0 = sound + n; int* rir = t0 + n; int* dsound = rir + n; int* drir = dsound + n;
    auto lookup = [&](int i, long long node_src) -> int {
        const long long scn = sc->scene[i];
        if (scn < 0 || scn >= sc->n_scenes) return -1;
        const long long dim = sc->index_dim[scn], r = sc->recv[i];
        if (r < 0 || r >= dim || node_src < 0 || node_src >= dim) return -1;
        const int base = sc->index_flat[sc->index_off[scn] + r * dim + node_src];
        if (base < 0) return -1;
        long long az = (-sc->rot[i]) % 360;
        if (az < 0) az += 360;
        return base + static_cast<int>(az / step);
    };
    int misses = 0;
    for (int i = 0; i < n; ++i) {                              // pass 1: everything except the audio_index advance
        const long long s = sc->sound[i];
        const bool silent = s < 0 || s >= n_src || sc->step_count[i] > sc->duration[i];
        sound[i] = silent ? 0 : static_cast<int>(s);
        dsound[i] = 0; drir[i] = -1;
        if (silent) { rir[i] = -1; t0[i] = 0; continue; }
        const int len = c.src_len[sound[i]];
        t0[i] = len == c.sr ? 0 : static_cast<int>(sc->audio_index[i]) * c.sr;
        rir[i] = lookup(i, sc->src[i]);
        bool miss = rir[i] < 0;
        if (dis) {
            const long long ds = sc->dis_sound[i];
            if (ds >= 0 && ds < n_src) {
                dsound[i] = static_cast<int>(ds);
                drir[i] = lookup(i, sc->dis_src[i]);
                miss = miss || drir[i] < 0;
            }
        }
        if (miss && miss_out && misses < n) miss_out[misses] = i;
        misses += miss;
    }
    if (misses) { *n_miss = misses; return 0; }
    for (int i = 0; i < n; ++i) {                              // pass 2: simulator.py:634-635
        if (rir[i] < 0) continue;
        const int len = c.src_len[sound[i]];
        if (len != c.sr && len >= c.sr) sc->audio_index[i] = (sc->audio_index[i] + 1) % (len / c.sr);
    }
    return 0;
}

// ---- request records of a multi-process vector env (ss_amd/deferred.py) -> unit columns ------------------------------
static long long find_key(const long long* keys, const long long* vals, int n, long long q) {
    if (!keys || n <= 0) return -1;
    const long long* p = std::lower_bound(keys, keys + n, q);
    return (p != keys + n && *p == q) ? vals[p - keys] : -1;
}

static int requests_to_units(ss_ctx* h, const long long* recs, int n, const ss_request_tables* tb, int* w, int* miss_out,
                             int* n_miss) {
    if (!h || !recs || !tb || n < 0 || !n_miss || tb->n_sounds < 0 || tb->n_tables < 0 || tb->n_pairs < 0) return SS_EINVAL;
    ssctx::Context& c = h->c;
    *n_miss = 0;
    const int n_src = static_cast<int>(c.src_len.size());
    int* sound = w; int* t0 = sound + n; int* rir = t0 + n; int* dsound = rir + n; int* drir = dsound + n;
    int misses = 0;
    auto slot_of = [&](long long table, long long recv, long long src) -> int {
        if (recv < 0 || src < 0 || recv >= (1 << 20) || src >= (1 << 20)) return -1;
        const long long s = find_key(tb->pair_keys, tb->pair_slots, tb->n_pairs, (table << 40) | (recv << 20) | src);
        if (s < 0 || (tb->stale && s < tb->n_slots && tb->stale[s])) return -1;
        if (tb->last_used && s < tb->n_slots) tb->last_used[s] = tb->tick;      // the caller's LRU clock
        return static_cast<int>(s);
    };
    for (int i = 0; i < n; ++i) {
        const long long* r = recs + static_cast<size_t>(i) * SS_REQ_WORDS;
        sound[i] = 0; t0[i] = 0; rir[i] = -1; dsound[i] = 0; drir[i] = -1;
        if (r[0] != 0) continue;                               // silent (simulator.py:610-612)
        const long long sid = find_key(tb->sound_keys, tb->sound_ids, tb->n_sounds, r[1]);
        const long long tid = find_key(tb->table_keys, tb->table_ids, tb->n_tables, r[3]);
        bool miss = sid < 0 || sid >= n_src || tid < 0 || r[2] < 0 || r[2] > 0x7fffffffLL;
        if (!miss) {
            sound[i] = static_cast<int>(sid);
            t0[i] = static_cast<int>(r[2]);
            rir[i] = slot_of(tid, r[4], r[5]);
            miss = rir[i] < 0;
            if (r[6] >= 0) {                                   // distractor (simulator.py:649-664): whole clip, own RIR
                const long long did = find_key(tb->sound_keys, tb->sound_ids, tb->n_sounds, r[6]);
                if (did < 0 || did >= n_src) miss = true;
                else {
                    dsound[i] = static_cast<int>(did);
                    drir[i] = slot_of(tid, r[4], r[7]);
                    miss = miss || drir[i] < 0;
                }
            }
        }
        if (miss && miss_out && misses < n) miss_out[misses] = i;
        misses += miss;
    }
    *n_miss = misses;
    return 0;
}

extern "C" {

int ss_ctx_requests_units(ss_ctx* h, const long long* recs, int n, const ss_request_tables* tb, int* units_out, int* miss_out,
                          int* n_miss) {
    if (!units_out) return SS_EINVAL;
    return requests_to_units(h, recs, n, tb, units_out, miss_out, n_miss);
}

int ss_ctx_observe_requests(ss_ctx* h, const long long* recs, int n, const ss_request_tables* tb, float* audiogoal,
                            float* spectrogram, int* miss_out, int* n_miss, void* stream) {
    if (!h || n < 0) return SS_EINVAL;
    std::vector<int>& w = h->c.sim_scratch;
    w.resize(static_cast<size_t>(n) * 5 + 1);
    const int rc = requests_to_units(h, recs, n, tb, w.data(), miss_out, n_miss);
    if (rc != 0 || n == 0 || *n_miss) return rc;
    ss_units u;
    std::memset(&u, 0, sizeof u);
    u.sound = w.data(); u.t0 = u.sound + n; u.rir = u.t0 + n;
    bool any_dis = false;
    for (int i = 0; i < n && !any_dis; ++i) any_dis = u.rir[n + n + i] >= 0;     // dis_rir column
    if (any_dis) { u.dis_sound = u.rir + n; u.dis_rir = u.dis_sound + n; }
    return ss_ctx_observe(h, &u, n, audiogoal, spectrogram, stream);
}

// k files -> k bank entries of the lent store (ss_miss_loader): entries off the free stack, then - with the recency arrays lent -
// the least recently used occupied ones (never one stamped with `tick`); files read into the pinned block, ONE scatter launch,
// the new rows' block spectra when `spectral`; loaded_slot / loaded_frames / evicted_slot report.  1 = not served, nothing changed.
static int loader_load_paths(ssctx::Context& c, ss_miss_loader* ld, const char* const* cp, int k, long long* last_used, long long tick,
                             int n_slots, bool spectral, hipStream_t st) {
    const int hb = (ld->cap + ssk::kB - 1) / ssk::kB;
    if (k <= 0 || k > ld->stage_rows || k > ld->loaded_cap) return 1;
    if (spectral && (!ld->stage_desc || c.h_blocks != hb || !c.hspec)) return 1;
    // entries: free ones first, then - if the caller lent its recency arrays - the least recently used occupied ones
    std::vector<int> victims;
    if (k > ld->n_free) {
        const int r = k - ld->n_free;
        if (!ld->used || !ld->use_seq || !ld->evicted_slot || !last_used || r > ld->evict_cap) return 1;
        std::vector<int> cand;
        for (int sl = 0; sl < n_slots; ++sl)
            if (ld->used[sl] && last_used[sl] < tick) cand.push_back(sl);     // (never a row this step resolves to)
        if (static_cast<int>(cand.size()) < r) return 1;       // the caller's own path raises its "store too small" there
        auto older = [&](int a, int b) {
            return last_used[a] != last_used[b] ? last_used[a] < last_used[b] : ld->use_seq[a] < ld->use_seq[b];
        };
        std::partial_sort(cand.begin(), cand.begin() + r, cand.end(), older);
        victims.assign(cand.begin(), cand.begin() + r);
    }
    std::vector<int> take(k);                                  // the entries the k new rows go to, in order
    for (int i = 0; i < k; ++i)
        take[i] = i < ld->n_free ? ld->free_slots[ld->n_free - 1 - i] : victims[i - ld->n_free];
    if (spectral) {                                            // (window offsets are int32 words from the bank's base)
        for (int i = 0; i < k; ++i)
            if ((static_cast<long long>(take[i]) + 1) * ld->bank_unit_stride >= (1LL << 31)) return 1;
    }
    if (c.miss_ev) {                                           // the staging block's previous scatter has run
        if (hipEventSynchronize(static_cast<hipEvent_t>(c.miss_ev)) != hipSuccess) return 1;
    } else {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return 1; }
        c.miss_ev = ev;
    }
    std::vector<int> kept(k), frames(k), status(k);
    sswav::read_many(cp, k, ld->stage, 2LL * ld->cap, ld->cap, ld->keep, false, kept.data(), frames.data(), status.data(),
                     ld->threads > 0 ? ld->threads : 1);
    for (int i = 0; i < k; ++i)
        if (status[i] != sswav::kOk && status[i] != sswav::kEmpty) return 1;     // scipy's semantics are the caller's reader's
#if !defined(SS_AB_NO_LOADER_LANE_ORDER)                       // (A/B builds: tests/test_wav_loader.py's write-after-read case fails without)
    if (order_behind_lanes(c, st)) return 1;                   // (overlap mode: steps in flight may read the entries rewritten below)
#endif
    // ---- commit
    for (size_t i = 0; i < victims.size(); ++i) ld->evicted_slot[i] = victims[i];
    ld->n_evicted = static_cast<int>(victims.size());
    const int from_free = k < ld->n_free ? k : ld->n_free;
    ld->n_free -= from_free;
    for (int i = 0; i < k; ++i) {
        const int slot = take[i];
        ld->stage_slot[i] = slot;
        ld->stage_len[i] = kept[i];
        ld->host_len[slot] = kept[i];
        ld->clipped[slot] = kept[i] < frames[i];
        if (ld->spec_stale) ld->spec_stale[slot] = 1;
        ld->loaded_slot[i] = slot; ld->loaded_frames[i] = frames[i];
        if (last_used && slot < n_slots) last_used[slot] = tick;
    }
    ld->n_loaded = k;
    int rc = ss_bank_scatter_rows_f32(ld->stage, 2LL * ld->cap, ld->stage_slot, ld->stage_len, k, ld->bank, ld->bank_unit_stride,
                                      ld->bank_chan_stride, ld->cap, ld->dev_len, st);
    if (rc == 0 && spectral) {
        // the new rows' block spectra H'_i = 2 rFFT(block i), straight into the spectral bank (scatter form of k_source_windows;
        // the descriptors are read from the pinned block in place)
        int w = 0;
        for (int i = 0; i < k; ++i)
            for (int ch = 0; ch < 2; ++ch)
                for (int b = 0; b < hb; ++b, ++w) {
                    const int slot = ld->stage_slot[i], left = ld->cap - b * ssk::kB;
                    int* d = ld->stage_desc + 5 * w;
                    d[0] = static_cast<int>(slot * ld->bank_unit_stride + static_cast<long long>(ch) * ld->bank_chan_stride + static_cast<long long>(b) * ssk::kB);
                    d[1] = left < ssk::kB ? left : ssk::kB;
                    d[2] = 0; d[3] = 0;
                    d[4] = (slot * 2 + ch) * hb + b;
                }
        ssk::SrcParams sp;
        rc = get_tables(&sp.tb);
        if (rc == 0) {
            sp.src = ld->bank;
            sp.desc = ld->stage_desc;
            sp.spec = reinterpret_cast<ssk::f32x4*>(const_cast<float*>(c.hspec));
            sp.desc_stride = 5;
            sp.scale = 1.0f;
            hipLaunchKernelGGL(ssk::k_source_windows, dim3(w), dim3(ssk::kT), 0, st, sp);
            rc = hip_err(hipGetLastError());
        }
        if (rc == 0 && ld->spec_stale)
            for (int i = 0; i < k; ++i) ld->spec_stale[ld->stage_slot[i]] = 0;
    }
    if (rc == 0) rc = hip_err(hipEventRecord(static_cast<hipEvent_t>(c.miss_ev), st));
    return rc;
}

// The miss path inside the call (include/ss_hip.h: ss_miss_loader).  Returns 1 = "not for the fast path" (nothing was changed).
static int serve_pose_misses(ss_ctx* h, const long long* recs, int n, ss_request_tables* tb, ss_miss_loader* ld, const int* miss,
                             int n_miss, hipStream_t st, bool spectral) {
    ssctx::Context& c = h->c;
    const int hb = (ld && ld->cap > 0) ? (ld->cap + ssk::kB - 1) / ssk::kB : 1;
    if (spectral && (!ld || !ld->stage_desc || c.h_blocks != hb)) return 1;
    if (!ld || !ld->table_dirs || !ld->pair_keys || !ld->pair_slots || !ld->free_slots || !ld->bank || !ld->dev_len ||
        !ld->host_len || !ld->clipped || !ld->stage || !ld->stage_slot || !ld->stage_len || !ld->loaded_key || !ld->loaded_slot ||
        !ld->loaded_frames || ld->cap < 2 || (ld->cap & 1) || tb->pair_keys != ld->pair_keys || tb->pair_slots != ld->pair_slots)
        return 1;
    const int n_src = static_cast<int>(c.src_len.size());
    std::vector<long long> keys;
    auto want = [&](long long tid, long long recv, long long src) -> bool {      // false: not serviceable here
        if (recv < 0 || src < 0 || recv >= (1 << 20) || src >= (1 << 20) || tid >= ld->n_table_dirs || !ld->table_dirs[tid]) return false;
        const long long key = (tid << 40) | (recv << 20) | src;
        const long long s = find_key(tb->pair_keys, tb->pair_slots, tb->n_pairs, key);
        if (s >= 0) return !(tb->stale && s < tb->n_slots && tb->stale[s]);      // resident (a stale row is the caller's reload)
        keys.push_back(key);
        return true;
    };
    for (int k = 0; k < n_miss; ++k) {
        const long long* r = recs + static_cast<size_t>(miss[k]) * SS_REQ_WORDS;
        const long long sid = find_key(tb->sound_keys, tb->sound_ids, tb->n_sounds, r[1]);
        const long long tid = find_key(tb->table_keys, tb->table_ids, tb->n_tables, r[3]);
        if (sid < 0 || sid >= n_src || tid < 0 || r[2] < 0 || r[2] > 0x7fffffffLL) return 1;
        if (!want(tid, r[4], r[5])) return 1;
        if (r[6] >= 0) {
            const long long did = find_key(tb->sound_keys, tb->sound_ids, tb->n_sounds, r[6]);
            if (did < 0 || did >= n_src || !want(tid, r[4], r[7])) return 1;
        }
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    const int k = static_cast<int>(keys.size());
    if (k == 0 || k > ld->stage_rows || k > ld->loaded_cap || tb->n_pairs + k > ld->pair_cap) return 1;
    int rc = 0;
    {
        std::vector<std::string> paths(k);
        std::vector<const char*> cp(k);
        for (int i = 0; i < k; ++i) {
            const long long key = keys[i];
            paths[i] = std::string(ld->table_dirs[key >> 40]) + "/" + std::to_string((key >> 20) & 0xFFFFF) + "_" +
                       std::to_string(key & 0xFFFFF) + ".wav";                   // simulator.py:615-616
            cp[i] = paths[i].c_str();
        }
        const int rc_l = loader_load_paths(c, ld, cp.data(), k, tb->last_used, tb->tick, tb->n_slots, spectral, st);
        if (rc_l == 1) return 1;
        for (int i = 0; i < k; ++i) ld->loaded_key[i] = keys[i];
        if (ld->n_evicted) {                                   // the evicted entries' pairs leave the sorted arrays (one compacting pass)
            std::vector<char> gone(static_cast<size_t>(tb->n_slots), 0);
            for (int i = 0; i < ld->n_evicted; ++i) gone[ld->evicted_slot[i]] = 1;
            int o = 0;
            for (int i = 0; i < tb->n_pairs; ++i) {
                const long long sl = ld->pair_slots[i];
                if (sl >= 0 && sl < tb->n_slots && gone[sl]) continue;
                ld->pair_keys[o] = ld->pair_keys[i]; ld->pair_slots[o] = ld->pair_slots[i]; ++o;
            }
            tb->n_pairs = o;
        }
        rc = rc_l;
    }
    // the sorted pair arrays, in place (keys ascending: merged from the back)
    {
        int i = tb->n_pairs - 1, j = k - 1, o = tb->n_pairs + k - 1;
        while (j >= 0) {
            if (i >= 0 && ld->pair_keys[i] > keys[j]) { ld->pair_keys[o] = ld->pair_keys[i]; ld->pair_slots[o] = ld->pair_slots[i]; --i; }
            else { ld->pair_keys[o] = keys[j]; ld->pair_slots[o] = ld->loaded_slot[j]; --j; }
            --o;
        }
        tb->n_pairs += k;
    }
    return rc;                                                 // (< 0: the rows are booked, the launch failed: the caller sees the error)
}

int ss_ctx_load_rir_files(ss_ctx* h, ss_miss_loader* ld, const char* const* paths, int k, long long* last_used, long long tick,
                          int n_slots, void* stream) {
    if (!h || !ld || !paths || k < 0) return SS_EINVAL;
    ld->n_loaded = 0; ld->n_evicted = 0;
    if (k == 0) return 0;
    ssctx::Context& c = h->c;
    if (!ld->free_slots || !ld->bank || !ld->dev_len || !ld->host_len || !ld->clipped || !ld->stage || !ld->stage_slot ||
        !ld->stage_len || !ld->loaded_slot || !ld->loaded_frames || ld->cap < 2 || (ld->cap & 1) || c.rir != ld->bank)
        return 1;
    // (rows of a store that keeps the spectral form get their block spectra right away: whichever form the next launch reads)
    return loader_load_paths(c, ld, paths, k, last_used, tick, n_slots, c.hspec != nullptr, static_cast<hipStream_t>(stream));
}

int ss_ctx_observe_requests_load(ss_ctx* h, const long long* recs, int n, ss_request_tables* tb, ss_miss_loader* ld, float* audiogoal,
                                 float* spectrogram, int* miss_out, int* n_miss, void* stream) {
    if (!h || n < 0 || !tb) return SS_EINVAL;
    if (ld) { ld->n_loaded = 0; ld->n_evicted = 0; }
    std::vector<int>& w = h->c.sim_scratch;
    w.resize(static_cast<size_t>(n) * 5 + 1);
    int rc = requests_to_units(h, recs, n, tb, w.data(), miss_out, n_miss);
    if (rc != 0 || n == 0) return rc;
    if (*n_miss) {
        ssctx::Context& c = h->c;
        // a launch that reads the SPECTRAL rows needs the new rows' block spectra first (built behind the scatter)
        bool any_dis = false;
        for (int i = 0; i < n && !any_dis; ++i) any_dis = recs[static_cast<size_t>(i) * SS_REQ_WORDS] == 0 && recs[static_cast<size_t>(i) * SS_REQ_WORDS + 6] >= 0;
        const bool spectral = c.hspec && !(c.spectral_max_units > 0 && c.rir && c.out_len <= ssk::kB && n > c.spectral_max_units && !any_dis);
        if (!ld || !miss_out || c.rir != ld->bank) return 0;
        std::vector<int> miss(miss_out, miss_out + (*n_miss < n ? *n_miss : n));
        rc = serve_pose_misses(h, recs, n, tb, ld, miss.data(), static_cast<int>(miss.size()), static_cast<hipStream_t>(stream), spectral);
        if (rc == 1) return 0;                                 // reported as ss_ctx_observe_requests reports it
        if (rc) return rc;
        rc = requests_to_units(h, recs, n, tb, w.data(), miss_out, n_miss);
        if (rc != 0 || *n_miss) return rc;
    }
    ss_units u;
    std::memset(&u, 0, sizeof u);
    u.sound = w.data(); u.t0 = u.sound + n; u.rir = u.t0 + n;
    bool any_dis = false;
    for (int i = 0; i < n && !any_dis; ++i) any_dis = u.rir[n + n + i] >= 0;
    if (any_dis) { u.dis_sound = u.rir + n; u.dis_rir = u.dis_sound + n; }
    return ss_ctx_observe(h, &u, n, audiogoal, spectrogram, stream);
}

}  // extern "C"

// ---- length-bucketed RIR bank ---------------------------------------------------------------------------
// buckets[0..n): HOST array; bucket b holds bank entries [first_b, first_b + n_entries_b) as planar rows of cap_b samples
// in an allocation of its own.  Bucket 0 fills the legacy fields of ConvParams, the others p.bk[].
static int fill_buckets(ssk::ConvParams& p, const ss_rir_bucket* bk, int n_buckets, bool spectral, int* nbh_max) {
    if (!bk || n_buckets < 1 || n_buckets > ssk::kMaxBuckets) return SS_EINVAL;
    int hb_max = 1;
    for (int b = 0; b < n_buckets; ++b) {
        if (!bk[b].rir || bk[b].cap < 2 || (bk[b].cap & 1) || bk[b].n_entries < 0 || bk[b].first < 0) return SS_EINVAL;
        if (b && bk[b].first < bk[b - 1].first + bk[b - 1].n_entries) return SS_EINVAL;     // ascending, disjoint ranges
        if (spectral && !bk[b].hspec) return SS_EINVAL;
        hb_max = std::max(hb_max, (bk[b].cap + ssk::kB - 1) / ssk::kB);
    }
    if (bk[0].first != 0) return SS_EINVAL;
    p.rir = bk[0].rir;
    p.rir_unit_stride = 2LL * bk[0].cap;
    p.rir_chan_stride = bk[0].cap;
    p.rir_elem_stride = 1;
    p.rir_cap = bk[0].cap;
    p.hspec = spectral ? reinterpret_cast<const ssk::f32x4*>(bk[0].hspec) : nullptr;
    p.h_blocks = spectral ? (bk[0].cap + ssk::kB - 1) / ssk::kB : 0;
    p.n_buckets = n_buckets;
    for (int b = 1; b < n_buckets; ++b)
        p.bk[b - 1] = ssk::BankBucket{bk[b].rir, reinterpret_cast<const ssk::f32x4*>(bk[b].hspec), bk[b].first, bk[b].cap,
                                      (bk[b].cap + ssk::kB - 1) / ssk::kB, 0};
    *nbh_max = hb_max;
    return 0;
}

static bool buckets_spectral(const ss_rir_bucket* bk, int n_buckets, int flags) {
    if (!bk || (flags & SS_FLAG_CROSSFADE)) return false;
    for (int b = 0; b < n_buckets; ++b) if (!bk[b].hspec) return false;
    return n_buckets > 0;
}

extern "C" {

int ss_fftconv_binaural_buckets_f32(const float* spec, const ss_rir_bucket* buckets, int n_buckets, const int* rir_len,
                                    const int* unit_desc, float* out, int n_units, int n_valid, int out_len, int flags,
                                    void* stream) {
    if (n_units == 0) return 0;
    if (!out || n_units < 0 || !buckets || n_buckets < 1) return SS_EINVAL;
    const bool spectral = buckets_spectral(buckets, n_buckets, flags);
    ssk::ConvParams p;
    int n_cus = 1, nbh_max = 1;
    int rc = fill_conv(p, &n_cus, spec, buckets[0].rir, rir_len, unit_desc, 2LL * buckets[0].cap, buckets[0].cap, 1,
                       buckets[0].cap, n_valid, out_len);
    if (rc == 0) rc = fill_buckets(p, buckets, n_buckets, spectral, &nbh_max);
    if (rc) return rc;
    p.out = out;
    const int nb_y = n_valid == 0 ? 1 : (n_valid + ssk::kB - 1) / ssk::kB;
    if (spectral) return launch_conv_spec<false>(p, n_units, nb_y, flags, static_cast<hipStream_t>(stream), n_cus);
    return launch_conv<false>(p, n_units, nb_y, flags, n_cus, static_cast<hipStream_t>(stream));
}

int ss_audio_obs_buckets_f32(const float* spec, const ss_rir_bucket* buckets, int n_buckets, const int* rir_len,
                             const int* unit_desc, float* audiogoal, float* spectrogram, int n_units, int n_valid,
                             int out_len, int pad_mode, int flags, void* stream) {
    if (n_units == 0) return 0;
    if (!spectrogram || n_units < 0 || !buckets || n_buckets < 1) return SS_EINVAL;
    if (pad_mode != SS_PAD_REFLECT && pad_mode != SS_PAD_CONSTANT) return SS_EINVAL;
    if (out_len < ssk::kNfft / 2 + 1) return SS_EINVAL;
    const bool spectral = buckets_spectral(buckets, n_buckets, flags);
    ssk::ConvParams p;
    int n_cus = 1, nbh_max = 1;
    int rc = fill_conv(p, &n_cus, spec, buckets[0].rir, rir_len, unit_desc, 2LL * buckets[0].cap, buckets[0].cap, 1,
                       buckets[0].cap, n_valid, out_len);
    if (rc == 0) rc = fill_buckets(p, buckets, n_buckets, spectral, &nbh_max);
    if (rc) return rc;
    p.pad_mode = pad_mode;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out_len <= ssk::kB && p.t4 <= 26) {
        p.out = audiogoal;
        p.sgram = spectrogram;
        return spectral ? launch_conv_spec<true>(p, n_units, 1, flags, st, n_cus) : launch_conv<true>(p, n_units, 1, flags, n_cus, st);
    }
    if (wide_one_block_ok(out_len, n_valid, flags, spectral)) {
        p.out = audiogoal;
        p.sgram = spectrogram;
        return launch_conv<true>(p, n_units, 1, flags, n_cus, st, true);
    }
    if (obs_rows_ok(out_len, n_valid, nbh_max, flags, spectral, audiogoal != nullptr)) {
        p.out = audiogoal;
        p.sgram = spectrogram;
        return spectral ? launch_obs_rows<true>(p, n_units, flags, n_cus, st) : launch_obs_rows<false>(p, n_units, flags, n_cus, st);
    }
    if (!audiogoal) return SS_EINVAL;                   // cross-faded long rows hand over through memory
    rc = ss_fftconv_binaural_buckets_f32(spec, buckets, n_buckets, rir_len, unit_desc, audiogoal, n_units, n_valid, out_len,
                                         flags, stream);
    if (rc) return rc;
    return spectrogram_of_rows(audiogoal, spectrogram, n_units, out_len, n_valid, pad_mode, stream);
}

// The context's bank as length buckets (replaces ss_ctx_set_rir_bank + ss_ctx_set_rir_spectra for such banks; borrowed
// device pointers, the array itself is copied).  Steps whose units all sit in bucket 0 keep the loop-free kernels.
int ss_ctx_set_rir_buckets(ss_ctx* h, const ss_rir_bucket* buckets, int n_buckets, const int* rir_len) {
    if (!h || !buckets || !rir_len || n_buckets < 1 || n_buckets > ssk::kMaxBuckets) return SS_EINVAL;
    ssk::ConvParams probe;
    int hb_max = 1;
    for (auto& b : probe.bk) b = ssk::BankBucket{nullptr, nullptr, 0x7fffffff, 0, 0, 0};
    int rc = fill_buckets(probe, buckets, n_buckets, false, &hb_max);
    if (rc) return rc;
    ssctx::Context& c = h->c;
    const int nbh_old = c.rir_cap > 0 ? ssctx::ceil_div(c.rir_cap, c.kb) : 1;
    c.buckets.assign(buckets, buckets + n_buckets);
    c.rir = buckets[0].rir; c.rir_len = rir_len; c.rir_us = 2LL * buckets[0].cap; c.rir_cs = buckets[0].cap; c.rir_es = 1;
    c.rir_cap = hb_max * c.kb;                                 // planning depth: the longest bucket's blocks
    c.hspec = nullptr; c.h_blocks = 0;
    if (hb_max != nbh_old) {                                   // the set of partition offsets per key changes
        const int nby = c.n_valid > 0 ? ssctx::ceil_div(c.n_valid, c.kb) : 1;
        c.stride = hb_max + nby - 1;
        ssctx::cache_reset(c);
        if (c.pool) {
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) return hip_err(e);
            (void)hipFree(c.pool);
            c.pool = nullptr;
            c.pool_entries = 0;
        }
    }
    return 0;
}

}  // extern "C"

// ---- RIR files -> staging rows (host only; see ss_wavio.hpp) ------------------------------------------------------
extern "C" int ss_wav_read_rirs_f32(const char* const* paths, int n, float* dst, long long row_stride, int cap, int keep,
                                    int planar, int* kept_out, int* frames_out, int* status_out, int n_threads) {
    if (n == 0) return 0;
    if (!paths || !dst || !kept_out || !frames_out || !status_out || n < 0 || cap < 1 || row_stride < 2LL * cap) return SS_EINVAL;
    sswav::read_many(paths, n, dst, row_stride, cap, keep, planar != 0, kept_out, frames_out, status_out, n_threads);
    return 0;
}

// Staged rows (wav layout, pinned host or device memory) -> planar bank rows + their lengths, one launch (k_scatter_rows)
extern "C" int ss_bank_scatter_rows_f32(const float* staged, long long staged_row_stride, const int* slots, const int* lens,
                                        int n, float* bank, long long unit_stride, int chan_stride, int cap, int* bank_len,
                                        void* stream) {
    if (n == 0) return 0;
    if (!staged || !slots || !lens || !bank || n < 0 || cap <= 0 || unit_stride <= 0 || chan_stride <= 0 ||
        staged_row_stride < 2LL * cap) return SS_EINVAL;
    // the kernel dereferences all three inputs: pageable host memory here would be a GPU memory fault (the process dies), not
    // an error code - refuse anything the runtime does not know as device-accessible (pinned / registered host, device, managed)
    for (const void* ptr : {static_cast<const void*>(staged), static_cast<const void*>(slots), static_cast<const void*>(lens)}) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) { (void)hipGetLastError(); return SS_EINVAL; }
        if (attr.type != hipMemoryTypeHost && attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged) return SS_EINVAL;
    }
    ssk::ScatterRowsParams p;
    p.staged = staged; p.slots = slots; p.lens = lens; p.bank = bank; p.bank_len = bank_len;
    p.staged_stride = staged_row_stride; p.unit_stride = unit_stride; p.chan_stride = chan_stride; p.cap = cap;
    for (int lo = 0; lo < n; lo += 65535) {                    // (grid.y limit)
        const int m = n - lo < 65535 ? n - lo : 65535;
        ssk::ScatterRowsParams q = p;
        q.staged += static_cast<size_t>(lo) * staged_row_stride; q.slots += lo; q.lens += lo;
        hipLaunchKernelGGL(ssk::k_scatter_rows, dim3((cap + 511) / 512, m), dim3(256), 0, static_cast<hipStream_t>(stream), q);
    }
    return hip_err(hipGetLastError());
}

extern "C" int ss_rows_gather_f32(const float* const* src, const int* n_floats, int n, float* dst, long long row_stride,
                                  int row_floats, int n_threads) {
    if (n == 0) return 0;
    if (!src || !n_floats || !dst || n < 0 || row_floats < 0 || row_stride < row_floats) return SS_EINVAL;
    sswav::gather_rows(src, n_floats, n, dst, row_stride, row_floats, n_threads);
    return 0;
}

// ss_context.hpp — host side of the self-contained entry points (include/ss_hip.h, "context API").
//
// What the reference does per env and per step in Python (soundspaces/simulator.py:608-666: pick the clip window,
// re-FFT the source inside scipy.signal.fftconvolve, convolve, slice) is split here into
//   * a PLANNER (pure host C++, no HIP): unit = {sound, t0, rir, ...} -> partition offsets, window-set keys, the
//     int32[8] unit descriptors the kernels take; one pass over the N units of a step, no Python per unit;
//   * a bounded CACHE of source-window spectra keyed (sound, t0, wrap): an LRU over fixed-size entries of a device
//     pool (SS1.0 keys repeat for ever -> all hits after warm-up; SS2.0 draws a new t0 per env and step -> the LRU
//     recycles entries instead of growing without bound);
//   * a pinned DESCRIPTOR RING: descriptors are written into page-locked memory and reach the device with one
//     asynchronous copy per step on the caller's stream (no pageable staging, no host sync).
// The planner mirrors sound-spaces_amd/ss_amd/planning.py (tests/test_context.py checks them against each other).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/ss_hip.h"

namespace ssctx {

constexpr int kDirectDescUnits = 256;   // steps up to this size read their descriptors in place from pinned memory
constexpr int kRing = 16;         // descriptor ring slots (steps in flight before a slot is reused)
constexpr int kGroup = 4;         // slots released per completion event
constexpr int kGuardTicks = 8;    // a cache entry used within the last kGuardTicks observe() calls is never evicted
                                  // (overlap mode: kRing - steps on the other lane may still be reading it; whatever is
                                  //  older than a full ring has finished, see the ring's completion events)
constexpr int kLanes = 4;         // internal streams of the overlap mode (ss_ctx_set_overlap): at most kGroup - the last
                                  // n_lanes ticks of a ring group are then on n_lanes different lanes, each lane's last

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }   // a >= 0, b > 0
inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : -static_cast<int>(e); }

struct WindowSet {
    int m_min = 0, count = 0;     // stored partition offsets m_min .. m_min+count-1 (count 0: nothing to convolve)
};

// planning.plan_window_set: offsets m in [-(nbh_max-1), nby-1] whose window x[t0+(m-1)KB : t0+(m+1)KB] is not all zero
inline WindowSet plan_window_set(int source_len, long long t0, int nbh_max, int nby, bool wrap, int kb) {
    const long long limit = wrap ? 2LL * source_len : source_len;
    WindowSet ws;
    bool any = false;
    for (int m = -(nbh_max - 1); m <= nby - 1; ++m) {
        if (t0 + (long long)(m + 1) * kb > 0 && t0 + (long long)(m - 1) * kb < limit) {
            if (!any) { ws.m_min = m; any = true; }
            ++ws.count;
        }
    }
    return ws;
}

struct Entry {
    uint64_t key = 0;
    int m_min = 0, count = 0;
    long long tick = -1;
    int prev = -1, next = -1;     // LRU list (head = most recent)
    bool used = false;
};

struct Context {
    // configuration
    int sr = 0, n_valid = 0, out_len = 0, pad_mode = 0, wrap_mode = 0, kb = 0, spec_floats = 0;
    int device = 0;
    // sources: one flat device buffer
    float* src_dev = nullptr;
    size_t src_cap = 0, src_used = 0;
    std::vector<int> src_off, src_len;
    // RIR bank (borrowed device pointers)
    const float* rir = nullptr;
    const int* rir_len = nullptr;
    long long rir_us = 0;
    int rir_cs = 0, rir_es = 1, rir_cap = 0;
    const float* hspec = nullptr; // spectral form of the same bank (optional, borrowed)
    int h_blocks = 0;
    std::vector<ss_rir_bucket> buckets;   // length-bucketed bank (ss_ctx_set_rir_buckets); empty: the single bank above
    // window-spectra cache
    int stride = 1;               // pool slots per entry = nbh_max + nby - 1
    int n_entries = 0;            // host bookkeeping (may run ahead of the device pool, see cache_grow)
    int pool_entries = 0;         // entries the device pool holds
    float* pool = nullptr;        // [pool_entries * stride][spec_floats]
    std::vector<Entry> entries;
    std::unordered_map<uint64_t, int> map;
    int lru_head = -1, lru_tail = -1;
    std::vector<int> free_list;
    long long tick = 0;
    long long hits = 0, misses = 0, evictions = 0, grows = 0;
    // descriptor ring
    int ring_cap = 0, ring_k = 0;                 // units per slot; next slot
    int* h_desc = nullptr;                        // pinned [kRing][ring_cap][8]
    int* d_desc = nullptr;                        // device  [kRing][ring_cap][8]
    int win_cap = 0;
    int* h_win = nullptr;                         // pinned [kRing][win_cap][5]: {src_offset, src_len, start, wrap, pool slot}
    int* d_win = nullptr;
    hipEvent_t ev_done[kRing / kGroup] = {};      // group g's launches have finished
    bool group_open = false;                      // a group has been started and its event not yet recorded
    int open_group = 0;
    hipStream_t group_stream = nullptr;           // ... on this stream
    bool ev_made = false;
    float* ag_scratch = nullptr;                  // hand-over buffer for rows longer than one block (44.1 kHz) when the
    size_t ag_cap = 0;                            // caller does not want the audiogoal itself
    std::vector<int> sim_scratch;                 // ss_ctx_observe_sims: unit columns of the step
    // scratch of the last plan
    std::vector<int> new_win;                     // 5 ints per new window
    std::vector<int> new_entries;                 // cache entries inserted by the last plan (rolled back if the step fails)
    bool plan_only_keys = false;                  // ss_ctx_plan inserted keys whose spectra were never computed
    hipStream_t last_stream = nullptr;            // stream of the last ss_ctx_observe that launched anything
    bool have_last_stream = false;
    hipEvent_t ev_xstream = nullptr;              // orders a new stream behind the previous one (pool reads / writes)
    // overlap mode (ss_ctx_set_overlap): consecutive steps alternate between internal streams ("lanes")
    int n_lanes = 1, lane_next = 0;
    void* miss_ev = nullptr;            // hipEvent_t behind the last scatter of ss_ctx_observe_requests_load (its staging block is reused)
    int spectral_max_units = 0;         // ss_ctx_set_spectral_policy: one-block rows take the spectral bank only for steps of <= this many units (0: always)
    int chip_share = 0;                 // ss_ctx_set_chip_share: launch sources the chip is shared with (0: the lane count)
    hipStream_t lane_stream[kLanes] = {};
    bool lane_dirty[kLanes] = {};                 // work issued on the lane since ev_lane was last recorded
    bool lane_joined[kLanes] = {};                // ev_lane holds a record ("everything issued on the lane so far")
    bool lane_join_valid[kLanes] = {};            // ... and lane_join_stream is the stream that waited for that record last
    hipStream_t lane_join_stream[kLanes] = {};
    hipEvent_t ev_lane[kLanes] = {};              // join: "everything issued on the lane so far"
    hipEvent_t ev_in = nullptr;                   // the caller's stream at the time of the call
    hipEvent_t ev_win[kLanes] = {};               // after the lane's latest k_source_windows launch
    long long win_seq[kLanes] = {};               // ... and how many it has recorded
    long long win_seen[kLanes][kLanes] = {};      // [waiting lane][recording lane]: highest seq already waited for
    hipEvent_t ev_done_l[kLanes][kRing / kGroup] = {};   // ring release, lane l's share of a group (l >= 1; lane 0 / single
                                                         // stream: ev_done)
    bool lanes_made = false;
};

inline uint64_t make_key(int sound, long long t0, bool wrap) {
    return (static_cast<uint64_t>(static_cast<uint32_t>(sound)) << 33) |
           (static_cast<uint64_t>(static_cast<uint32_t>(static_cast<int32_t>(t0))) << 1) | (wrap ? 1u : 0u);
}

inline void lru_unlink(Context& c, int e) {
    Entry& x = c.entries[e];
    if (x.prev >= 0) c.entries[x.prev].next = x.next; else c.lru_head = x.next;
    if (x.next >= 0) c.entries[x.next].prev = x.prev; else c.lru_tail = x.prev;
    x.prev = x.next = -1;
}
inline void lru_push_front(Context& c, int e) {
    Entry& x = c.entries[e];
    x.prev = -1;
    x.next = c.lru_head;
    if (c.lru_head >= 0) c.entries[c.lru_head].prev = e;
    c.lru_head = e;
    if (c.lru_tail < 0) c.lru_tail = e;
}

inline void cache_reset(Context& c) {
    c.map.clear();
    c.entries.assign(c.n_entries, Entry{});
    c.free_list.clear();
    for (int e = c.n_entries - 1; e >= 0; --e) c.free_list.push_back(e);
    c.lru_head = c.lru_tail = -1;
}

// Every entry is in use by recent steps: double the number of entries.  Only the host bookkeeping grows here (slot
// indices stay valid); the device pool is reallocated by observe() before anything is written to the new slots.
inline void cache_grow(Context& c) {
    const int old_n = c.n_entries;
    c.n_entries = old_n * 2;
    c.entries.resize(c.n_entries);
    for (int e = c.n_entries - 1; e >= old_n; --e) c.free_list.push_back(e);
    ++c.grows;
}

// Entry for (sound, t0, wrap): hit -> most recent; miss -> a free or evicted entry (growing the cache when every entry
// was used within the guard window), its windows appended to new_win.
inline int cache_get(Context& c, int sound, long long t0, bool wrap, int nbh_max, int nby) {
    const uint64_t key = make_key(sound, t0, wrap);
    auto it = c.map.find(key);
    if (it != c.map.end()) {
        const int e = it->second;
        if (c.entries[e].tick != c.tick) { ++c.hits; c.entries[e].tick = c.tick; lru_unlink(c, e); lru_push_front(c, e); }
        return e;
    }
    int e;
    if (!c.free_list.empty()) {
        e = c.free_list.back();
        c.free_list.pop_back();
    } else {
        e = c.lru_tail;
        if (e < 0 || c.entries[e].tick > c.tick - (c.n_lanes > 1 ? kRing : kGuardTicks)) {
            cache_grow(c);
            e = c.free_list.back();
            c.free_list.pop_back();
        } else {
            lru_unlink(c, e);
            c.map.erase(c.entries[e].key);
            ++c.evictions;
        }
    }
    ++c.misses;
    Entry& x = c.entries[e];
    const WindowSet ws = plan_window_set(c.src_len[sound], t0, nbh_max, nby, wrap, c.kb);
    x.key = key; x.m_min = ws.m_min; x.count = ws.count; x.tick = c.tick; x.used = true;
    lru_push_front(c, e);
    c.map.emplace(key, e);
    c.new_entries.push_back(e);
    for (int k = 0; k < ws.count; ++k) {
        const long long start = t0 + (long long)(ws.m_min + k - 1) * c.kb;
        const int row[5] = {c.src_off[sound], c.src_len[sound], static_cast<int>(start), wrap ? 1 : 0, e * c.stride + k};
        c.new_win.insert(c.new_win.end(), row, row + 5);
    }
    return e;
}

// A step that fails after its plan (allocation, copy, launch) never computes the spectra of the keys the plan inserted:
// take them out again, or the next plan would report them as hits on pool slots that hold nothing.
// (An entry evicted to make room is simply gone: its key re-plans as a miss.)
inline void cache_rollback(Context& c) {
    for (int e : c.new_entries) {
        Entry& x = c.entries[e];
        if (!x.used) continue;
        auto it = c.map.find(x.key);
        if (it != c.map.end() && it->second == e) c.map.erase(it);
        lru_unlink(c, e);
        x = Entry{};
        c.free_list.push_back(e);
    }
    c.new_entries.clear();
    c.new_win.clear();
}

struct PlanResult {
    int flags = 0;
    int n_new_windows = 0;
};

// Fill desc[n][8] for the units of one step.  Pure host code.
inline int plan_units(Context& c, const ss_units* u, int n, int* desc, PlanResult* res) {
    if (!u || n < 0 || (n > 0 && (!u->sound || !u->t0 || !u->rir))) return SS_EINVAL;
    const int nbh_max = c.rir_cap > 0 ? ceil_div(c.rir_cap, c.kb) : 1;
    const int nby = c.n_valid > 0 ? ceil_div(c.n_valid, c.kb) : 1;
    const int n_src = static_cast<int>(c.src_len.size());
    // Pass 1: every reason to refuse the step, BEFORE the cache is touched (planning is transactional: a refused step
    // leaves no keys behind).  Window sets are a pure function of (clip length, t0), so "does this term convolve anything"
    // needs no cache entry.
    {
        bool fade = false, dis_term = false;
        for (int i = 0; i < n; ++i) {
            if (u->rir[i] < 0) continue;
            const int s = u->sound[i];
            if (s < 0 || s >= n_src) return SS_EINVAL;
            const long long t0 = u->t0[i];
            const bool over = t0 + c.n_valid > c.src_len[s];
            const bool w0 = c.wrap_mode && over && (!u->wrap || u->wrap[i]);
            if (plan_window_set(c.src_len[s], t0, nbh_max, nby, w0, c.kb).count <= 0) continue;
            const int last = u->last_rir ? u->last_rir[i] : -1;
            const int dis = u->dis_rir ? u->dis_rir[i] : -1;
            if (last >= 0) {
                if (dis >= 0) return SS_EINVAL;
                fade = true;
            } else if (dis >= 0) {
                if (!u->dis_sound) return SS_EINVAL;
                const int ds = u->dis_sound[i];
                if (ds < 0 || ds >= n_src) return SS_EINVAL;
                if (plan_window_set(c.src_len[ds], 0, nbh_max, nby, false, c.kb).count > 0) dis_term = true;
            }
        }
        if (fade && dis_term) return SS_EINVAL;                   // a launch is either cross-faded or has distractors
    }
    ++c.tick;
    c.new_win.clear();
    c.new_entries.clear();
    bool any_dis = false, any_fade = false;
    int max_rir = -1;                                            // highest bank index a term of this step reads
    for (int i = 0; i < n; ++i) {
        int* d = desc + 8 * i;
        d[0] = -1; d[1] = d[2] = d[3] = 0; d[4] = -1; d[5] = d[6] = d[7] = 0;
        if (u->rir[i] < 0) continue;                              // silent unit / no RIR: exact zeros
        const int s = u->sound[i];
        if (s < 0 || s >= n_src) return SS_EINVAL;
        const long long t0 = u->t0[i];
        const bool over = t0 + c.n_valid > c.src_len[s];          // only windows past the clip end differ when wrapped
        const bool w0 = c.wrap_mode && over && (!u->wrap || u->wrap[i]);
        const int e0 = cache_get(c, s, t0, w0, nbh_max, nby);
        const Entry& x0 = c.entries[e0];
        if (x0.count <= 0) continue;                              // nothing of the clip under this window: silent
        d[0] = u->rir[i]; d[1] = e0 * c.stride; d[2] = x0.m_min; d[3] = x0.count;
        if (d[0] > max_rir) max_rir = d[0];
        const int last = u->last_rir ? u->last_rir[i] : -1;
        const int dis = u->dis_rir ? u->dis_rir[i] : -1;
        if (last >= 0) {
            if (dis >= 0) return SS_EINVAL;                       // term 1 is either a distractor or the previous RIR
            const bool w1 = c.wrap_mode && over && (u->last_wrap ? u->last_wrap[i] != 0 : (!u->wrap || u->wrap[i]));
            const int e1 = cache_get(c, s, t0, w1, nbh_max, nby);
            const Entry& x1 = c.entries[e1];
            d[4] = last; d[5] = e1 * c.stride; d[6] = x1.m_min; d[7] = x1.count;
            if (last > max_rir) max_rir = last;
            any_fade = true;
        } else if (dis >= 0) {
            if (!u->dis_sound) return SS_EINVAL;
            const int ds = u->dis_sound[i];
            if (ds < 0 || ds >= n_src) return SS_EINVAL;
            const int e1 = cache_get(c, ds, 0, false, nbh_max, nby);     // whole clip from its start (simulator.py:659-664)
            const Entry& x1 = c.entries[e1];
            if (x1.count > 0) {
                d[4] = dis; d[5] = e1 * c.stride; d[6] = x1.m_min; d[7] = x1.count; any_dis = true;
                if (dis > max_rir) max_rir = dis;
            }
        }
    }
    if (any_fade && any_dis) return SS_EINVAL;                    // a launch is either cross-faded or has distractors
    res->flags = any_fade ? SS_FLAG_CROSSFADE : (any_dis ? 0 : SS_FLAG_NO_DISTRACTOR);
    if (c.buckets.size() > 1 && max_rir < c.buckets[1].first) res->flags |= SS_FLAG_FIRST_BUCKET;
    res->n_new_windows = static_cast<int>(c.new_win.size() / 5);
    return 0;
}

}  // namespace ssctx

// ss_wavio.hpp — the RIR miss path's file reader (host C++, no HIP): many float32 binaural wav files straight into a
// staging block, on a pool of plain threads (no interpreter lock, no per-file array objects, no transposes).
//
// Reference: SoundSpacesSim._compute_audiogoal reads `<binaural_rir_dir>/<azimuth>/<recv>_<src>.wav` with
// scipy.io.wavfile.read on EVERY cache-missing step (soundspaces/simulator.py:615-618, "# float32"), from a data set of
// 867 GB (soundspaces/README.md:9); an unreadable file (ValueError) and an empty one become the zero RIR (:619-624).
// Here: a file is parsed once (RIFF header, "fmt " and "data" chunks), its first `keep` frames are read() directly into
// the caller's row - wav-interleaved [frames][2], the layout the file already has - and the row is zero-filled behind
// them.  Anything that is not a plain little-endian IEEE-float32 stereo RIFF/WAVE file is NOT interpreted here: it is
// reported (status) and the caller routes that one file through the Python reader, which reproduces scipy's behaviour
// (integer PCM, RIFX, malformed chunks -> ValueError -> zero RIR) exactly.
#pragma once
#include <fcntl.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace sswav {

enum Status : int {
    kOk = 0,            // row filled
    kUnsupported = 1,   // not a plain float32 stereo RIFF/WAVE file: the caller's Python reader decides (scipy semantics)
    kEmpty = 2,         // readable, zero frames            -> zero RIR (simulator.py:622-624)
    kMissing = 3,       // open() failed                    -> the caller raises / maps to the zero RIR (lenient)
    kTooLong = 4,       // more frames to keep than the row holds: nothing read, the caller grows its rows and retries
};

// A small persistent pool: the per-step pose misses are a handful of files, and starting threads for every call costs more
// than reading them (tens of microseconds per thread; far more under a sandboxed kernel).  Workers are created on first
// use and park on a condition variable; run(n, fn) calls fn(i) for i in [0, n) on up to `threads` of them plus the caller
// and returns when all are done.  One job at a time (callers are serialised by a mutex).
class Pool {
public:
    // One pool per PROCESS: a child of fork() inherits `workers_` without the threads behind it - run() would wait
    // for helpers that do not exist, ~Pool would join them.  The pool remembers the pid that built it; a process with another
    // pid gets a fresh one (the inherited object is abandoned, never destroyed: its mutexes may have been held at the fork).
    // Pools are not destroyed at exit either: the parked workers end with the process.
    static Pool& get() {
        static std::atomic<Pool*> inst{nullptr};
        static std::atomic_flag making = ATOMIC_FLAG_INIT;
        static std::once_flag fork_once;
        std::call_once(fork_once, [] { ::pthread_atfork(nullptr, nullptr, [] { making.clear(); }); });
        const pid_t me = ::getpid();
        Pool* p = inst.load(std::memory_order_acquire);
        if (p && p->pid_ == me) return *p;
        while (making.test_and_set(std::memory_order_acquire)) std::this_thread::yield();
        p = inst.load(std::memory_order_acquire);
        if (!p || p->pid_ != me) { p = new Pool(); p->pid_ = me; inst.store(p, std::memory_order_release); }
        making.clear(std::memory_order_release);
        return *p;
    }
    void run(int n, int threads, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        int helpers = (threads < 1 ? 1 : threads) - 1;
        if (helpers > n - 1) helpers = n - 1;
        if (helpers > kMax) helpers = kMax;
        if (helpers <= 0) { for (int i = 0; i < n; ++i) fn(i); return; }
        std::lock_guard<std::mutex> job_lock(job_mu_);
        {
            std::lock_guard<std::mutex> lk(mu_);
            while (static_cast<int>(workers_.size()) < helpers) workers_.emplace_back([this] { loop(); });
            fn_ = &fn; n_ = n; next_.store(0); pending_ = helpers; wanted_ = helpers; ++epoch_;
        }
        cv_.notify_all();
        for (;;) { const int i = next_.fetch_add(1); if (i >= n) break; fn(i); }
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    static constexpr int kMax = 15;
    Pool() = default;
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)>* fn;
            int n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && wanted_ > 0); });
                if (stop_) return;
                seen = epoch_;
                --wanted_;                                   // this worker joins the job (at most `helpers` do)
                fn = fn_; n = n_;
            }
            for (;;) { const int i = next_.fetch_add(1); if (i >= n) break; (*fn)(i); }
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_cv_.notify_one();
            }
        }
    }
    std::mutex job_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, pending_ = 0, wanted_ = 0;
    unsigned long long epoch_ = 0;
    bool stop_ = false;
    pid_t pid_ = 0;
};

inline bool read_exact(int fd, void* buf, size_t n) {
    char* p = static_cast<char*>(buf);
    while (n) {
        const ssize_t r = ::read(fd, p, n);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        if (r == 0) return false;
        p += r; n -= static_cast<size_t>(r);
    }
    return true;
}
inline uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
inline uint16_t le16(const unsigned char* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }

// One file -> one row.  dst: [cap][2] floats (interleaved) or, planar, dst[c * cap + n].  keep < 0: the whole file.
// frames_out: frames the FILE holds (the caller learns whether the row was clipped); returns the Status, *kept = frames stored.
// Four system calls per file - open, one read of the first 4 KiB (RIFF header, "fmt ", any LIST chunk and the start of the
// samples), one read of the rest straight into the row, close: a step that loads a handful of new poses is bound by the
// latency of these calls, not by the bytes (the first version parsed the header with one read per field: nine calls).
constexpr int kHeadBytes = 4096;
inline int read_one(const char* path, float* dst, int cap, int keep, bool planar, int* kept, int* frames_out,
                    std::vector<float>& scratch) {
    *kept = 0; *frames_out = 0;
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return kMissing;
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    alignas(8) unsigned char head[kHeadBytes];
    size_t have = 0;
    for (;;) {                                                  // (a short read is legal: keep reading until 4 KiB or EOF)
        const ssize_t r = ::read(fd, head + have, kHeadBytes - have);
        if (r < 0) { if (errno == EINTR) continue; return kUnsupported; }
        if (r == 0) break;
        have += static_cast<size_t>(r);
        if (have == kHeadBytes) break;
    }
    if (have < 12 || std::memcmp(head, "RIFF", 4) != 0 || std::memcmp(head + 8, "WAVE", 4) != 0) return kUnsupported;
    bool have_fmt = false;
    int channels = 0, bits = 0, tag = 0, block_align = 0;
    size_t pos = 12;
    for (;;) {
        if (pos + 8 > have) return kUnsupported;               // no data chunk within the first 4 KiB / truncated header
        const unsigned char* ch = head + pos;
        const uint32_t sz = le32(ch + 4);
        pos += 8;
        if (std::memcmp(ch, "fmt ", 4) == 0) {
            if (sz < 16 || sz > 40 || pos + sz > have) return kUnsupported;
            const unsigned char* f = head + pos;
            tag = le16(f); channels = le16(f + 2); block_align = le16(f + 12); bits = le16(f + 14);
            if (tag == 0xFFFE && sz >= 26) tag = le16(f + 24);    // WAVE_FORMAT_EXTENSIBLE: the sub-format's first word
            have_fmt = true;
            pos += sz + (sz & 1);
        } else if (std::memcmp(ch, "data", 4) == 0) {
            if (!have_fmt || tag != 3 || bits != 32 || channels != 2 || block_align != 8) return kUnsupported;
            const int frames = static_cast<int>(sz / 8);
            *frames_out = frames;
            if (frames == 0) return kEmpty;
            const int n = keep >= 0 && keep < frames ? keep : frames;
            if (n > cap) {
                //  only a file that HOLDS the frames its header claims may make the caller grow every row of its
                // bank: a truncated / corrupt one goes to the Python reader first (scipy reads it leniently or it becomes the
                // zero RIR, simulator.py:617-621)
                struct stat st;
                if (::fstat(fd, &st) != 0 || static_cast<uint64_t>(st.st_size) < pos + static_cast<uint64_t>(n) * 8) return kUnsupported;
                return kTooLong;
            }
            const size_t want = static_cast<size_t>(n) * 8;                          // bytes of samples to store
            const size_t in_head = have - pos < want ? have - pos : want;            // ... of which the first read holds
            float* tgt = dst;
            if (planar) { scratch.resize(static_cast<size_t>(n) * 2); tgt = scratch.data(); }
            std::memcpy(tgt, head + pos, in_head);
            if (in_head < want) {
                if (have < static_cast<size_t>(kHeadBytes)) return kUnsupported;      // EOF inside the data chunk: truncated
                if (!read_exact(fd, reinterpret_cast<char*>(tgt) + in_head, want - in_head)) return kUnsupported;
            }
            if (n < frames) {                                     // a clipped read: the bytes asked for arrived, but a file cut
                struct stat st;                                   // short BEHIND them is not what scipy would have accepted
                if (::fstat(fd, &st) != 0 || static_cast<uint64_t>(st.st_size) < pos + sz) return kUnsupported;
            }
            if (!planar) {
                std::memset(dst + static_cast<size_t>(n) * 2, 0, static_cast<size_t>(cap - n) * 8);
            } else {
                for (int c = 0; c < 2; ++c) {
                    float* row = dst + static_cast<size_t>(c) * cap;
                    for (int i = 0; i < n; ++i) row[i] = scratch[2 * static_cast<size_t>(i) + c];
                    std::memset(row + n, 0, static_cast<size_t>(cap - n) * 4);
                }
            }
            *kept = n;
            return kOk;
        } else {                                                   // LIST, fact, ...: skipped (word-aligned)
            pos += static_cast<size_t>(sz) + (sz & 1);
        }
    }
}

// n files -> n rows of `row_stride` floats each, on up to n_threads threads (files are dealt out dynamically).
// Rows whose file did not load (status != kOk) are zero-filled, kept = 0.
inline void read_many(const char* const* paths, int n, float* dst, long long row_stride, int cap, int keep, bool planar,
                      int* kept, int* frames, int* status, int n_threads) {
    Pool::get().run(n, n_threads, [&](int i) {
        thread_local std::vector<float> scratch;
        float* row = dst + static_cast<size_t>(i) * static_cast<size_t>(row_stride);
        status[i] = read_one(paths[i], row, cap, keep, planar, &kept[i], &frames[i], scratch);
        if (status[i] != kOk) std::memset(row, 0, static_cast<size_t>(cap) * 8);
    });
}

// n host arrays -> n rows of a staging block: row i = src[i][0 .. n_floats[i]) followed by zeros up to row_floats.
// (SoundSpaces 2.0 hands every env a new RIR every step, continuous_simulator.py:419: the trainer gathers the step's live
// RIRs into ONE pinned block - these copies, off the interpreter lock and on several threads, were a fifth of that step.)
inline void gather_rows(const float* const* src, const int* n_floats, int n, float* dst, long long row_stride, int row_floats,
                        int n_threads) {
    Pool::get().run(n, n_threads, [&](int i) {
        float* row = dst + static_cast<size_t>(i) * static_cast<size_t>(row_stride);
        const int k = n_floats[i] < row_floats ? (n_floats[i] > 0 ? n_floats[i] : 0) : row_floats;
        if (k > 0) std::memcpy(row, src[i], static_cast<size_t>(k) * 4);
        std::memset(row + k, 0, static_cast<size_t>(row_floats - k) * 4);
    });
}

}  // namespace sswav

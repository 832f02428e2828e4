// ss_hip.hip — host side of libss_hip.so: table construction and kernel launches (gfx950 only).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/ss_hip.h"
#include "ss_kernels.hpp"
#include "ss_tables.hpp"

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

template <bool FUSE>
int launch_conv(const ssk::ConvParams& p, int n_units, int nb_y, int flags, int n_cus, hipStream_t st) {
    if (nb_y < 1 || nb_y > 3 || (FUSE && nb_y != 1)) return SS_EINVAL;
    const bool simple = (flags & SS_FLAG_NO_DISTRACTOR) && !(flags & SS_FLAG_CROSSFADE) && nb_y == 1 && p.rir_cap <= ssk::kB;
    if ((flags & SS_FLAG_CROSSFADE) && (p.fade_len < 1 || p.fade_len > 2 * ssk::kPrevPairs - 2)) return SS_EINVAL;
    const dim3 grid(2 * n_units, nb_y), block(ssk::kT);
    // more rows than CUs: persistent workgroups that prefetch the next row's RIR under the current row's FFT passes
    const bool planar = p.rir_elem_stride == 1 && !(p.rir_cap & 1) && !(reinterpret_cast<size_t>(p.rir) & 7) &&
                        !(p.rir_unit_stride & 1) && !(p.rir_chan_stride & 1) && p.rir_cap >= 2;
    static const bool no_rows = getenv("SS_HIP_NO_ROW_KERNEL") != nullptr;          // A/B switch for benchmarking
    if constexpr (!FUSE) {              // the fused kernel gains nothing from it (measured), see k_conv_rows
        if (simple && planar && !no_rows && 2 * n_units > n_cus) {
            hipLaunchKernelGGL(ssk::k_conv_rows, dim3(n_cus), block, 0, st, p, 2 * n_units);
            return hip_err(hipGetLastError());
        }
    }
    if (flags & SS_FLAG_CROSSFADE) hipLaunchKernelGGL((ssk::k_conv<FUSE, false, true>), grid, block, 0, st, p);
    else if (simple) hipLaunchKernelGGL((ssk::k_conv<FUSE, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((ssk::k_conv<FUSE, false>), grid, block, 0, st, p);
    return hip_err(hipGetLastError());
}

}  // namespace

extern "C" {

int ss_block_len(void) { return ssk::kB; }
int ss_spec_floats(void) { return 2 * ssk::kSpecComplex; }
int ss_version(void) { return 1; }

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
    hipLaunchKernelGGL(ssk::k_source_windows, dim3(n_windows), dim3(ssk::kT), 0,
                       static_cast<hipStream_t>(stream), p);
    return hip_err(hipGetLastError());
}

static int fill_conv(ssk::ConvParams& p, int* n_cus, const float* spec, const float* rir, const int* rir_len,
                     const int* unit_desc, long long us, int cs, int es, int cap, int n_valid, int out_len) {
    if (!spec || !rir || !rir_len || !unit_desc) return SS_EINVAL;
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

int ss_spectrogram_f32(const float* x, float* out, int n_units, int len, int pad_mode, void* stream) {
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
    if (!audiogoal) return SS_EINVAL;                   // long rows hand over through HBM/L2
    rc = ss_fftconv_binaural_f32(spec, rir, rir_len, unit_desc, audiogoal, n_units, rir_unit_stride,
                                 rir_chan_stride, rir_elem_stride, rir_cap, n_valid, out_len, flags, stream);
    if (rc) return rc;
    return ss_spectrogram_f32(audiogoal, spectrogram, n_units, out_len, pad_mode, stream);
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

}  // extern "C"

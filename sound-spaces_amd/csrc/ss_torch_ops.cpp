// ss_torch_ops.cpp — the torch custom-op layer of SURVEY 8(b) as a TORCH_LIBRARY extension (host C++ only; the kernels
// live in libss_hip.so, which this file calls through the C ABI of include/ss_hip.h).
//
// Until round 4 every torch.ops.ss_hip.* op was a torch.library registration in Python over ctypes (ss_amd/ops.py): an
// eager observation - the reference's own call chain, one env per call (soundspaces/tasks/nav.py:102-105 ->
// soundspaces/simulator.py:690-701, 608-666) - paid ~13 us of Python planner and ~15 us of ctypes + argument checks on top
// of a 21-us batch-1 launch.  The hot ops are defined HERE, with the planner folded in through the context API:
//
//   ss_hip::spectrogram        x [N,2,n]                                -> [N,65,T4,2]      (nav.py:86-100)
//   ss_hip::audio_obs          pre-planned descriptors                  -> (audiogoal, spectrogram)
//   ss_hip::ctx_observe        unit columns (CPU int32 tensors)         -> spectrogram rows written in place
//   ss_hip::eager_obs          ONE unit as scalars: ss_ctx_observe (C++ planner, window cache, launch) + one async copy of
//                              both outputs into pinned memory + stream synchronise - an eager observation is ONE dispatch
//   ss_hip::ctx_register / ctx_unregister   the raw ss_ctx* behind an AudioContext.handle
//
//   round 5: source_windows, fftconv_binaural, rir_spectra, fftconv_binaural_spec, audio_obs_spec, audio_features, intensity
//
// Schemas are those ss_amd/ops.py registered before (it now only adds the ops not defined here: the two stand-alone feature
// kernels and the Meta shape functions).  Built in-tree by
// sound-spaces_amd/build.py (g++ against the torch headers; links libss_hip.so by $ORIGIN).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // PyTorch-ROCm: HIP devices carry the device type "cuda"
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#include <torch/library.h>

#include <mutex>
#include <unordered_map>

#include "../../include/ss_hip.h"

namespace {

std::mutex g_mu;
std::unordered_map<int64_t, ss_ctx*> g_ctx;

ss_ctx* ctx_of(int64_t handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(handle);
    TORCH_CHECK(it != g_ctx.end(), "ss_hip: unknown context handle ", handle);
    return it->second;
}

void check_rc(int rc, const char* what) {
    if (rc == 0) return;
    if (rc == SS_EINVAL) TORCH_CHECK(false, what, " failed: invalid argument");
    TORCH_CHECK(false, what, " failed: hipError_t ", -rc);
}

const float* fptr(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), name, " must live on the GPU (this path has no CPU implementation)");
    TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous(), name, ": expected contiguous float32");
    return t.data_ptr<float>();
}
const int* iptr(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), name, " must live on the GPU");
    TORCH_CHECK(t.scalar_type() == at::kInt && t.is_contiguous(), name, ": expected contiguous int32");
    return t.data_ptr<int>();
}

int64_t t4_of(int64_t n) { return ((1 + n / 160) + 3) / 4; }

void* stream_of(const at::Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

at::Tensor spectrogram(const at::Tensor& x, int64_t pad_mode) {
    TORCH_CHECK(x.dim() == 3 && x.size(1) == 2, "x must be [N, 2, n]");
    const float* xp = fptr(x, "x");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    at::Tensor out = at::empty({x.size(0), 65, t4_of(x.size(2)), 2}, x.options());
    check_rc(ss_spectrogram_f32(xp, out.data_ptr<float>(), static_cast<int>(x.size(0)), static_cast<int>(x.size(2)),
                                static_cast<int>(pad_mode), stream_of(x)), "ss_spectrogram_f32");
    return out;
}

std::tuple<at::Tensor, at::Tensor> audio_obs(const at::Tensor& spec, const at::Tensor& rir_bank, const at::Tensor& rir_len,
                                             const at::Tensor& unit_desc, int64_t n_valid, int64_t out_len, int64_t pad_mode,
                                             bool interleaved, int64_t flags) {
    TORCH_CHECK(rir_bank.dim() == 3, "rir_bank must be [R,2,L] (planar) or [R,L,2] (wav-interleaved)");
    TORCH_CHECK(unit_desc.dim() == 2 && unit_desc.size(1) == 8, "unit_desc must be [N, 8]");
    const int64_t cap = interleaved ? rir_bank.size(1) : rir_bank.size(2);
    const long long us = 2 * cap;
    const int cs = interleaved ? 1 : static_cast<int>(cap), es = interleaved ? 2 : 1;
    const int64_t N = unit_desc.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(spec.device());
    at::Tensor ag = at::empty({N, 2, out_len}, spec.options());
    at::Tensor sg = at::empty({N, 65, t4_of(out_len), 2}, spec.options());
    check_rc(ss_audio_obs_f32(fptr(spec, "spec"), fptr(rir_bank, "rir_bank"), iptr(rir_len, "rir_len"), iptr(unit_desc, "unit_desc"),
                              ag.data_ptr<float>(), sg.data_ptr<float>(), static_cast<int>(N), us, cs, es, static_cast<int>(cap),
                              static_cast<int>(n_valid), static_cast<int>(out_len), static_cast<int>(pad_mode),
                              static_cast<int>(flags), stream_of(spec)), "ss_audio_obs_f32");
    return {ag, sg};
}

// ---- round 5: the rest of the op layer ("no hot op crosses ctypes") ---------------------------------
constexpr int64_t kSpecFloats = 2 * 16384, kKB = 16384;

at::Tensor source_windows(const at::Tensor& src, const at::Tensor& win_desc) {
    TORCH_CHECK(win_desc.dim() == 2 && win_desc.size(1) == 4, "win_desc must be [W, 4]");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(src.device());
    at::Tensor out = at::empty({win_desc.size(0), kSpecFloats}, src.options());
    check_rc(ss_source_windows_f32(fptr(src, "src"), iptr(win_desc, "win_desc"), out.data_ptr<float>(),
                                   static_cast<int>(win_desc.size(0)), stream_of(src)), "ss_source_windows_f32");
    return out;
}

at::Tensor fftconv_binaural(const at::Tensor& spec, const at::Tensor& rir_bank, const at::Tensor& rir_len,
                            const at::Tensor& unit_desc, int64_t n_valid, int64_t out_len, bool interleaved, int64_t flags) {
    TORCH_CHECK(rir_bank.dim() == 3, "rir_bank must be [R,2,L] (planar) or [R,L,2] (wav-interleaved)");
    TORCH_CHECK(unit_desc.dim() == 2 && unit_desc.size(1) == 8, "unit_desc must be [N, 8]");
    const int64_t cap = interleaved ? rir_bank.size(1) : rir_bank.size(2), N = unit_desc.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(spec.device());
    at::Tensor out = at::empty({N, 2, out_len}, spec.options());
    check_rc(ss_fftconv_binaural_f32(fptr(spec, "spec"), fptr(rir_bank, "rir_bank"), iptr(rir_len, "rir_len"),
                                     iptr(unit_desc, "unit_desc"), out.data_ptr<float>(), static_cast<int>(N), 2 * cap,
                                     interleaved ? 1 : static_cast<int>(cap), interleaved ? 2 : 1, static_cast<int>(cap),
                                     static_cast<int>(n_valid), static_cast<int>(out_len), static_cast<int>(flags),
                                     stream_of(spec)), "ss_fftconv_binaural_f32");
    return out;
}

at::Tensor rir_spectra(const at::Tensor& rir_bank) {
    TORCH_CHECK(rir_bank.dim() == 3 && rir_bank.size(1) == 2, "rir_bank must be planar [R, 2, cap]");
    const int64_t R = rir_bank.size(0), cap = rir_bank.size(2), hb = (cap + kKB - 1) / kKB;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(rir_bank.device());
    at::Tensor out = at::empty({R, 2, hb, kSpecFloats}, rir_bank.options());
    check_rc(ss_rir_spectra_f32(fptr(rir_bank, "rir_bank"), out.data_ptr<float>(), static_cast<int>(R), 2 * cap,
                                static_cast<int>(cap), static_cast<int>(cap), stream_of(rir_bank)), "ss_rir_spectra_f32");
    return out;
}

at::Tensor fftconv_binaural_spec(const at::Tensor& spec, const at::Tensor& hspec, const at::Tensor& rir_len,
                                 const at::Tensor& unit_desc, int64_t n_valid, int64_t out_len, int64_t flags) {
    TORCH_CHECK(hspec.dim() == 4 && unit_desc.dim() == 2 && unit_desc.size(1) == 8, "hspec [R,2,hb,F], unit_desc [N,8]");
    const int64_t N = unit_desc.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(spec.device());
    at::Tensor out = at::empty({N, 2, out_len}, spec.options());
    check_rc(ss_fftconv_binaural_spec_f32(fptr(spec, "spec"), fptr(hspec, "hspec"), iptr(rir_len, "rir_len"),
                                          iptr(unit_desc, "unit_desc"), out.data_ptr<float>(), static_cast<int>(N),
                                          static_cast<int>(hspec.size(2)), static_cast<int>(n_valid), static_cast<int>(out_len),
                                          static_cast<int>(flags), stream_of(spec)), "ss_fftconv_binaural_spec_f32");
    return out;
}

std::tuple<at::Tensor, at::Tensor> audio_obs_spec(const at::Tensor& spec, const at::Tensor& hspec, const at::Tensor& rir_len,
                                                  const at::Tensor& unit_desc, int64_t n_valid, int64_t out_len, int64_t pad_mode,
                                                  int64_t flags) {
    TORCH_CHECK(hspec.dim() == 4 && unit_desc.dim() == 2 && unit_desc.size(1) == 8, "hspec [R,2,hb,F], unit_desc [N,8]");
    const int64_t N = unit_desc.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(spec.device());
    at::Tensor ag = at::empty({N, 2, out_len}, spec.options());
    at::Tensor sg = at::empty({N, 65, t4_of(out_len), 2}, spec.options());
    check_rc(ss_audio_obs_spec_f32(fptr(spec, "spec"), fptr(hspec, "hspec"), iptr(rir_len, "rir_len"), iptr(unit_desc, "unit_desc"),
                                   ag.data_ptr<float>(), sg.data_ptr<float>(), static_cast<int>(N), static_cast<int>(hspec.size(2)),
                                   static_cast<int>(n_valid), static_cast<int>(out_len), static_cast<int>(pad_mode),
                                   static_cast<int>(flags), stream_of(spec)), "ss_audio_obs_spec_f32");
    return {ag, sg};
}

// log-mel + GCC-PHAT of x [N, 2, n] in ONE pass (k_features; BASELINE configs[4]'s fused sensor)
std::tuple<at::Tensor, at::Tensor> audio_features(const at::Tensor& x, const at::Tensor& mel_start, const at::Tensor& mel_w,
                                                  double mel_eps, int64_t max_lag, double gcc_eps, int64_t pad_mode) {
    TORCH_CHECK(x.dim() == 3 && x.size(1) == 2, "x must be [N, 2, n]");
    TORCH_CHECK(mel_w.dim() == 2 && mel_start.dim() == 1 && mel_start.size(0) == mel_w.size(0), "mel_start [n_mels], mel_w [n_mels, max_len]");
    const int64_t N = x.size(0), n = x.size(2), T = 1 + n / 160;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    at::Tensor lm = at::empty({N, mel_w.size(0), T, 2}, x.options());
    at::Tensor gc = at::empty({N, 2 * max_lag + 1, T}, x.options());
    check_rc(ss_audio_features_f32(fptr(x, "x"), static_cast<int>(N), static_cast<int>(n), static_cast<int>(pad_mode), nullptr,
                                   lm.data_ptr<float>(), iptr(mel_start, "mel_start"), fptr(mel_w, "mel_w"),
                                   static_cast<int>(mel_w.size(0)), static_cast<int>(mel_w.size(1)), static_cast<float>(mel_eps),
                                   gc.data_ptr<float>(), static_cast<int>(max_lag), static_cast<float>(gcc_eps), stream_of(x)),
             "ss_audio_features_f32");
    return {lm, gc};
}

at::Tensor intensity(const at::Tensor& audiogoal, int64_t num_frame) {
    TORCH_CHECK(audiogoal.dim() == 3 && audiogoal.size(1) == 2, "audiogoal must be [N, 2, n]");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(audiogoal.device());
    at::Tensor out = at::empty({audiogoal.size(0)}, audiogoal.options());
    check_rc(ss_intensity_f32(fptr(audiogoal, "audiogoal"), out.data_ptr<float>(), static_cast<int>(audiogoal.size(0)),
                              static_cast<int>(audiogoal.size(2)), static_cast<int>(num_frame), stream_of(audiogoal)),
             "ss_intensity_f32");
    return out;
}

void ctx_register(int64_t handle, int64_t ptr) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_ctx[handle] = reinterpret_cast<ss_ctx*>(static_cast<uintptr_t>(ptr));
}
void ctx_unregister(int64_t handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_ctx.erase(handle);
}

const int* cpu_i32(const at::Tensor& t, const char* name, int64_t n) {
    TORCH_CHECK(!t.is_cuda() && t.scalar_type() == at::kInt && t.is_contiguous() && t.numel() == n, name,
                ": expected a contiguous CPU int32 tensor of ", n, " entries");
    return t.data_ptr<int>();
}

at::Tensor ctx_observe(int64_t handle, const at::Tensor& sound, const at::Tensor& t0, const at::Tensor& rir,
                       at::Tensor spectrogram) {
    const int64_t n = sound.numel();
    TORCH_CHECK(spectrogram.dim() == 4 && spectrogram.size(0) == n, "spectrogram must be [n, 65, T4, 2]");
    float* sg = const_cast<float*>(fptr(spectrogram, "spectrogram"));
    ss_units u{};
    u.sound = cpu_i32(sound, "sound", n);
    u.t0 = cpu_i32(t0, "t0", n);
    u.rir = cpu_i32(rir, "rir", n);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(spectrogram.device());
    check_rc(ss_ctx_observe(ctx_of(handle), &u, static_cast<int>(n), nullptr, sg, stream_of(spectrogram)), "ss_ctx_observe");
    return spectrogram;
}

// One eager observation (batch 1): plan + launch + ONE device-to-host copy of [audiogoal 2*sr | spectrogram 65*T4*2] into
// pinned memory + stream synchronise.  dev / host are flat float32 buffers of at least 2*sr (+ 65*T4*2) elements.
void eager_obs(int64_t handle, int64_t sound, int64_t t0, int64_t rir, int64_t dis_sound, int64_t dis_rir, int64_t last_rir,
               int64_t wrap, int64_t last_wrap, at::Tensor dev, at::Tensor host, int64_t sr, bool want_spectrogram,
               bool want_audiogoal) {
    // (pinned memory is the caller's contract: Tensor::is_pinned() asks the driver on every call, ~5 us of a 50-us observation)
    TORCH_CHECK(!host.is_cuda() && host.scalar_type() == at::kFloat && host.is_contiguous(),
                "host: expected a (pinned) contiguous float32 CPU tensor");
    TORCH_CHECK(dev.is_cuda(), "dev must live on the GPU (it names the device, and holds the outputs unless it is empty)");
    const int64_t n_ag = 2 * sr, n_sg = want_spectrogram ? 65 * t4_of(sr) * 2 : 0;
    // dev empty: the kernels write straight into the pinned host buffer (pinned memory is device-mapped under ROCm: the
    // 141 KB cross PCIe as the kernel's own stores instead of as a copy-engine job behind it)
    const bool direct = dev.numel() == 0;
    TORCH_CHECK(host.numel() >= n_ag + n_sg && (direct || dev.numel() >= n_ag + n_sg), "eager_obs: buffers too small");
    float* d = direct ? host.data_ptr<float>() : const_cast<float*>(fptr(dev, "dev"));
    int s = static_cast<int>(sound), t = static_cast<int>(t0), r = static_cast<int>(rir), ds = static_cast<int>(dis_sound),
        dr = static_cast<int>(dis_rir), lr = static_cast<int>(last_rir);
    unsigned char w = wrap != 0, lw = last_wrap != 0;
    ss_units u{};
    u.sound = &s; u.t0 = &t; u.rir = &r;
    if (dr >= 0) { u.dis_sound = &ds; u.dis_rir = &dr; }
    if (lr >= 0) { u.last_rir = &lr; u.last_wrap = &lw; }
    u.wrap = &w;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev.device());
    hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.device().index()).stream();
    // want_audiogoal false (a task with a SpectrogramSensor only): the waveform never leaves the CU - no 128 KB over PCIe
    TORCH_CHECK(want_audiogoal || want_spectrogram, "eager_obs: nothing to compute");
    check_rc(ss_ctx_observe(ctx_of(handle), &u, 1, want_audiogoal ? d : nullptr, want_spectrogram ? d + n_ag : nullptr, st),
             "ss_ctx_observe");
    check_rc(ss_ctx_join(ctx_of(handle), st), "ss_ctx_join");    // (a context on overlap lanes: the read-back waits for the lane)
    hipError_t e = hipSuccess;
    if (!direct) {
        const int64_t lo = want_audiogoal ? 0 : n_ag;
        e = hipMemcpyAsync(host.data_ptr<float>() + lo, d + lo, sizeof(float) * static_cast<size_t>(n_ag + n_sg - lo),
                           hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    TORCH_CHECK(e == hipSuccess, "eager_obs: ", hipGetErrorString(e));
}

}  // namespace

TORCH_LIBRARY(ss_hip, m) {
    m.def("spectrogram(Tensor x, int pad_mode=0) -> Tensor");
    m.def("audio_obs(Tensor spec, Tensor rir_bank, Tensor rir_len, Tensor unit_desc, int n_valid, int out_len, "
          "int pad_mode=0, bool interleaved=False, int flags=0) -> (Tensor, Tensor)");
    m.def("ctx_observe(int ctx, Tensor sound, Tensor t0, Tensor rir, Tensor(a!) spectrogram) -> Tensor(a!)");
    m.def("ctx_register(int handle, int ptr) -> ()", &ctx_register);
    m.def("ctx_unregister(int handle) -> ()", &ctx_unregister);
    m.def("eager_obs(int ctx, int sound, int t0, int rir, int dis_sound, int dis_rir, int last_rir, int wrap, int last_wrap, "
          "Tensor(a!) dev, Tensor(b!) host, int sr, bool want_spectrogram, bool want_audiogoal=True) -> ()");
    m.def("source_windows(Tensor src, Tensor win_desc) -> Tensor");
    m.def("fftconv_binaural(Tensor spec, Tensor rir_bank, Tensor rir_len, Tensor unit_desc, int n_valid, int out_len, "
          "bool interleaved=False, int flags=0) -> Tensor");
    m.def("rir_spectra(Tensor rir_bank) -> Tensor");
    m.def("fftconv_binaural_spec(Tensor spec, Tensor hspec, Tensor rir_len, Tensor unit_desc, int n_valid, int out_len, "
          "int flags=0) -> Tensor");
    m.def("audio_obs_spec(Tensor spec, Tensor hspec, Tensor rir_len, Tensor unit_desc, int n_valid, int out_len, "
          "int pad_mode=0, int flags=0) -> (Tensor, Tensor)");
    m.def("audio_features(Tensor x, Tensor mel_start, Tensor mel_w, float mel_eps=1e-6, int max_lag=32, float gcc_eps=1e-8, "
          "int pad_mode=0) -> (Tensor, Tensor)");
    m.def("intensity(Tensor audiogoal, int num_frame=150) -> Tensor");
    m.def("native_ops() -> int", []() -> int64_t { return 2; });   // 2: + the ops of round 5 (ss_amd/ops.py registers the rest)
}

TORCH_LIBRARY_IMPL(ss_hip, CUDA, m) {
    m.impl("spectrogram", &spectrogram);
    m.impl("audio_obs", &audio_obs);
    m.impl("source_windows", &source_windows);
    m.impl("fftconv_binaural", &fftconv_binaural);
    m.impl("rir_spectra", &rir_spectra);
    m.impl("fftconv_binaural_spec", &fftconv_binaural_spec);
    m.impl("audio_obs_spec", &audio_obs_spec);
    m.impl("audio_features", &audio_features);
    m.impl("intensity", &intensity);
}

TORCH_LIBRARY_IMPL(ss_hip, CompositeExplicitAutograd, m) {
    m.impl("ctx_observe", &ctx_observe);
    m.impl("eager_obs", &eager_obs);
}

// calib_traffic.hip — FETCH_SIZE / WRITE_SIZE calibration for the access patterns of libss_hip.so (MI355X_MICROARCH: "calibrate
// on a known byte count in your own access pattern").  Every kernel moves EXACTLY `bytes` through one pattern over a buffer
// of 1 GiB (4 x the Infinity Cache), three launches each; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
// (separate passes) and divide: factor = bytes / (counter x 1024).  scripts/gpu_profile_r4.sh does that and stores the
// factors in profiles/r4/traffic.json.
//   rd8_nt    8 B/lane  __builtin_nontemporal_load    the RIR rows of k_conv / k_obs_rows (ld_stream<c32>)
//   rd16      16 B/lane plain loads                   window spectra, stash read-back, spectral bank
//   rd16_nt   16 B/lane nontemporal loads             spectral bank rows (ld_stream<f32x4>)
//   wr16_nt   16 B/lane nontemporal stores            block-spectra stash of k_obs_rows
//   wr8_nt    8 B/lane  nontemporal stores            audiogoal rows (st_stream<c32>)
//   wr4       4 B/lane  strided stores                spectrogram rows (every other float of a channel-last row)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void rd8_nt(const f2* p, float* sink, size_t n) {
    f2 acc = {0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(p + i);
    if (acc.x == 12345.f) sink[0] = acc.y;
}
__global__ void rd16(const f4* p, float* sink, size_t n) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc.x == 12345.f) sink[0] = acc.y + acc.z + acc.w;
}
__global__ void rd16_nt(const f4* p, float* sink, size_t n) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(p + i);
    if (acc.x == 12345.f) sink[0] = acc.y + acc.z + acc.w;
}
__global__ void wr16_nt(f4* p, size_t n) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, p + i);
}
__global__ void wr8_nt(f2* p, size_t n) {
    const f2 v = {1.f, 2.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, p + i);
}
__global__ void wr4(float* p, size_t n) {          // every other float: n stores over 2 n floats
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[2 * i] = 1.f;
}

int main() {
    const size_t bytes = 1ull << 30;
    void* buf = nullptr; float* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    const dim3 grid(256 * 8), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rd8_nt, grid, block, 0, 0, (const f2*)buf, sink, bytes / 8);
        hipLaunchKernelGGL(rd16, grid, block, 0, 0, (const f4*)buf, sink, bytes / 16);
        hipLaunchKernelGGL(rd16_nt, grid, block, 0, 0, (const f4*)buf, sink, bytes / 16);
        hipLaunchKernelGGL(wr16_nt, grid, block, 0, 0, (f4*)buf, bytes / 16);
        hipLaunchKernelGGL(wr8_nt, grid, block, 0, 0, (f2*)buf, bytes / 8);
        hipLaunchKernelGGL(wr4, grid, block, 0, 0, (float*)buf, bytes / 8);      // bytes/8 stores of 4 B = bytes/2 written, bytes spanned
    }
    hipDeviceSynchronize();
    std::printf("calib: %zu bytes per read / nt-write launch; wr4 stores %zu bytes over a span of %zu\n", bytes, bytes / 2, bytes);
    return 0;
}

// LDS throughput per CU on gfx950 for the access shapes of the FFT passes: 1024 threads, conflict-free lane-consecutive
// addresses, ds_{read,write}_b{64,128}.  Prints bytes per clock and CU (clock from s_memtime-free wall time x an assumed rate is
// avoided: the kernel counts its own cycles with s_memrealtime? no - clock64()).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[32768];        // 128 KiB
    const int t = threadIdx.x;
    f2* p2 = reinterpret_cast<f2*>(lds) + t;                           // 8-byte lane stride
    f4* p4 = reinterpret_cast<f4*>(lds) + t;                           // 16-byte lane stride
    f2 a2 = {float(t), 1.f}; f4 a4 = {float(t), 1.f, 2.f, 3.f};
    for (int i = t; i < 32768; i += 1024) lds[i] = float(i);
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { f2 v = p2[1024 * u]; a2 += v; }                        // ds_read_b64: 16 x 8 KiB = 128 KiB per sweep
            if (MODE == 1) { p2[1024 * u] = a2; a2.x += 1.f; }                       // ds_write_b64
            if (MODE == 2 && u < 8) { f4 v = p4[1024 * u]; a4 += v; }                // ds_read_b128: 8 x 16 KiB = 128 KiB
            if (MODE == 3 && u < 8) { p4[1024 * u] = a4; a4.x += 1.f; }              // ds_write_b128
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    const long long c1 = clock64();
    if (t == 0) cyc[blockIdx.x] = c1 - c0;
    out[blockIdx.x * 1024 + t] = a2.x + a2.y + a4.x + a4.y + a4.z + a4.w + lds[t];
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&cyc, 8 * 256);
    const int iters = 2000;
    const char* names[4] = {"ds_read_b64", "ds_write_b64", "ds_read_b128", "ds_write_b128"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, out, cyc, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, out, cyc, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, out, cyc, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
            const double bytes = 131072.0 * iters;
            printf("%-14s %.3f ms  clock64 delta %lld  -> %.1f B per clock64 tick and CU; %.1f GB/s per CU (wall)\n", names[mode], ms,
                   h[0], bytes / (double)h[0], bytes / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}

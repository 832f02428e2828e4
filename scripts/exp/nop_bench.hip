// Does the `s_nop 0` hipcc puts between dependent inline-asm VALU blocks on gfx942/gfx950 cost time?
// (LLVM GCNHazardRecognizer: "assume inline asm has dst forwarding hazard".)  Two kernels with the SAME packed-f32 instruction
// stream - four interleaved dependent chains per lane, a radix-2-butterfly-like mix - once as one asm statement per instruction
// (compiler inserts the nops) and once as ONE asm block (no nops).  1024 threads per workgroup, one workgroup per CU, like k_conv.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float c32 __attribute__((ext_vector_type(2)));
#define ADD(r, a, b) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b))
#define MUL(r, a, b) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b))
#define FMA(r, a, b, c) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c))
template <int DEP>
__global__ __launch_bounds__(1024) void k_sep(const c32* x, c32* y, int iters) {
    c32 a = x[threadIdx.x], b = x[threadIdx.x + 1024], c = x[threadIdx.x + 2048], d = x[threadIdx.x + 3072], w = x[threadIdx.x + 4096];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (DEP) {          // every instruction consumes the result of the one before it
                ADD(a, a, w); MUL(a, a, w); FMA(a, a, w, b); ADD(a, a, c);
            } else {            // four independent chains, round-robin: producer and consumer three instructions apart
                ADD(a, a, w); ADD(b, b, w); ADD(c, c, w); ADD(d, d, w);
                MUL(a, a, w); MUL(b, b, w); FMA(c, c, w, c); FMA(d, d, w, d);
            }
        }
    }
    y[blockIdx.x * 1024 + threadIdx.x] = a + b + c + d;
}
template <int DEP>
__global__ __launch_bounds__(1024) void k_one(const c32* x, c32* y, int iters) {
    c32 a = x[threadIdx.x], b = x[threadIdx.x + 1024], c = x[threadIdx.x + 2048], d = x[threadIdx.x + 3072], w = x[threadIdx.x + 4096];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (DEP)
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_fma_f32 %0, %0, %4, %1\n v_pk_add_f32 %0, %0, %2"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w));
            else
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w));
        }
    }
    y[blockIdx.x * 1024 + threadIdx.x] = a + b + c + d;
}
template <class F>
float time_it(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 100.f;   // us per launch
}
int main() {
    c32 *x, *y; hipMalloc(&x, 8 * 5120); hipMalloc(&y, 8 * 1024 * 1024); hipMemset(x, 0, 8 * 5120);
    const int iters = 200;
    for (int wgs : {256, 512}) {
        for (int threads : {256, 512, 1024}) {
            float a = time_it([&] { hipLaunchKernelGGL(k_sep<1>, dim3(wgs), dim3(threads), 0, 0, x, y, iters); });
            float b = time_it([&] { hipLaunchKernelGGL(k_one<1>, dim3(wgs), dim3(threads), 0, 0, x, y, iters); });
            float c = time_it([&] { hipLaunchKernelGGL(k_sep<0>, dim3(wgs), dim3(threads), 0, 0, x, y, iters); });
            float d = time_it([&] { hipLaunchKernelGGL(k_one<0>, dim3(wgs), dim3(threads), 0, 0, x, y, iters); });
            printf("wgs %d threads %d (waves/SIMD %d): dependent chain sep %.1f us one-block %.1f us | 4 chains sep %.1f one-block %.1f\n",
                   wgs, threads, threads / 256, a, b, c, d);
        }
    }
    return 0;
}

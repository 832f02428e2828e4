// hip_shim.h — TEST INFRASTRUCTURE: just enough of the HIP device vocabulary to compile the kernel
// headers of sound-spaces_amd/csrc for the HOST (clang++, for ext_vector_type), so their index algebra can be
// checked against the oracle without a GPU.  Threads of a workgroup run as cooperative fibers (hostsim.cpp);
// __syncthreads() / wave_sync() are real barriers over the workgroup's / the 64-lane wave's fibers.
// Never part of the product; the product only runs the gfx950 build.
#pragma once
#define SSK_HOSTSIM 1
#include <cmath>
#include <cstddef>
#include <cstdint>

struct dim3 { unsigned x = 1, y = 1, z = 1; };
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

void hostsim_syncthreads();
void hostsim_wave_sync();
float hostsim_lane_read(float v, int src_lane);   // value of `v` held by lane src_lane of the caller's wave
#define __syncthreads() hostsim_syncthreads()
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int atomicMax(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }   // fibers run one at a time

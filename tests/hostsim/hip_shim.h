// hip_shim.h — TEST INFRASTRUCTURE: just enough of the HIP device vocabulary to compile the kernel
// headers of sound-spaces_amd/csrc for the HOST, so their index algebra can be checked against the
// oracle without a GPU.  Threads of a workgroup run as cooperative fibers (hostsim.cpp); __syncthreads()
// yields to the scheduler.  Never part of the product; the product only runs the gfx950 build.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct dim3 { unsigned x = 1, y = 1, z = 1; };
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

void hostsim_syncthreads();
#define __syncthreads() hostsim_syncthreads()
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }

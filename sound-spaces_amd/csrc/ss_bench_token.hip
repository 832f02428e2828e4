// ss_bench_token.hip — BENCH-ONLY helper (libss_bench.so; not part of libss_hip.so, nothing in ss_amd/ loads it).
//
// In the reference, step k+1's observation cannot be asked for before the policy has turned step k's observation into an
// action (ss_baselines/av_nav/ppo/ppo_trainer.py:133-150: actor_critic.act on rollouts.observations[step], then envs.step).
// bench.py's dependent-step mode needs a stand-in for that consumer: ONE workgroup that reads the observation slot the step
// has just written (a strided sample of it, first and last element included) and writes a token; the next step is ordered
// behind it on the caller's stream.  The kernel does no useful arithmetic - its only job is to BE the data dependency.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void k_policy_token(const float* __restrict__ obs, long long n, float* __restrict__ token) {
    __shared__ float part[256];
    const int t = threadIdx.x;
    float acc = 0.f;
    if (n > 0) {
        const long long stride = n / 2048 > 0 ? n / 2048 : 1;
        for (int k = 0; k < 8; ++k) {
            const long long i = (static_cast<long long>(k) * 256 + t) * stride;
            acc += obs[i < n ? i : n - 1];
        }
        if (t == 255) acc += obs[n - 1];
    }
    part[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) part[t] += part[t + s];
        __syncthreads();
    }
    if (t == 0) token[0] = part[0];
}

extern "C" int ssb_policy_token(const float* obs, long long n_floats, float* token, void* stream) {
    hipLaunchKernelGGL(k_policy_token, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), obs, n_floats, token);
    return static_cast<int>(hipGetLastError());
}

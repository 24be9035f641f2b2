// Per-step reward-event bookkeeping of the rollout driver (agent_environment.py:66-76) as two
// small launches (one when a single workgroup covers the rows) instead of a dozen tensor ops,
// and for EVERY env instead of the first two rows.
//
// The reference classifies a step by comparing float32 rewards with constants: any reward >= 1
// is a delivery, == 0.3 a plating, == 0.1 an onion put into a pot (rows 0..1 only, whatever the
// agent count).  Kept literally for those rows; the all-env totals use the same comparisons per
// env.  Rewards are sums of 0.1, 0.3 and whole numbers, so round(10 r) is exact and the totals
// are integers: the result does not depend on the order of summation.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oc {

constexpr int EVENT_BLOCK = 256;

int reward_event_chunks(int N) {
    int c = (N + 2047) / 2048;
    return c < 1 ? 1 : c > 256 ? 256 : c;
}

// rows 0..1 the way the reference counts them: rewards[0:2].mean() in float32, accumulated as a
// Python float (agent_environment.py:76), and the three event flags of those rows
__device__ __forceinline__ void first_rows_update(const float* __restrict__ rewards, int N, double* __restrict__ first_rows) {
    const int n = N < 2 ? N : 2;
    float sum = 0.0f;
    bool d = false, p = false, o = false;
    for (int j = 0; j < n; ++j) {
        const float r = rewards[j];
        sum = __fadd_rn(sum, r);
        d |= r >= 1.0f; p |= r == 0.3f; o |= r == 0.1f;
    }
    first_rows[0] += (double)__fdiv_rn(sum, (float)n);
    first_rows[1] += d ? 1.0 : 0.0;
    first_rows[2] += p ? 1.0 : 0.0;
    first_rows[3] += o ? 1.0 : 0.0;
}

// pass 1: block c scans envs c, c + C, ... and writes its four sums to partial[c][4].  Launched
// as a single block with `totals` set it is the whole job: the sums go straight to the totals
// and lane 4 does the first-rows part.
__global__ void __launch_bounds__(EVENT_BLOCK)
reward_events_partial_kernel(const float* __restrict__ rewards, int E, int A, uint64_t* __restrict__ partial,
                             uint64_t* __restrict__ totals, double* __restrict__ first_rows) {
    uint32_t tenths = 0, deliv = 0, plate = 0, pot = 0;  // per lane: at most E/blocks envs, far below 2^32
    for (int e = blockIdx.x * EVENT_BLOCK + threadIdx.x; e < E; e += gridDim.x * EVENT_BLOCK) {
        bool d = false, p = false, o = false;
        for (int a = 0; a < A; ++a) {
            const float r = rewards[(size_t)e * A + a];
            tenths += (uint32_t)(r * 10.0f + 0.5f);
            d |= r >= 1.0f;
            p |= r == 0.3f;
            o |= r == 0.1f;
        }
        deliv += d; plate += p; pot += o;
    }
    uint64_t v[4] = {tenths, deliv, plate, pot};
    __shared__ uint64_t part[EVENT_BLOCK / 64][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) part[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint64_t s = 0;
        for (int w = 0; w < EVENT_BLOCK / 64; ++w) s += part[w][threadIdx.x];
        if (totals) totals[threadIdx.x] += s;
        else partial[(size_t)blockIdx.x * 4 + threadIdx.x] = s;
    } else if (threadIdx.x == 4 && totals && first_rows) {
        first_rows_update(rewards, E * A, first_rows);
    }
}

// pass 2: wave k (of four) adds the chunk sums of total k to the running totals; the last lane of
// the block does the first-rows part
__global__ void __launch_bounds__(256)
reward_events_finish_kernel(const float* __restrict__ rewards, int N, const uint64_t* __restrict__ partial, int C,
                            uint64_t* __restrict__ totals, double* __restrict__ first_rows) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint64_t s = 0;
    for (int c = lane; c < C; c += 64) s += partial[(size_t)c * 4 + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) totals[k] += s;
    if (threadIdx.x == 255 && first_rows) first_rows_update(rewards, N, first_rows);
}

hipError_t launch_reward_events(const float* rewards, int E, int A, uint64_t* partial, uint64_t* totals,
                                double* first_rows, hipStream_t stream) {
    const int C = reward_event_chunks(E * A);
    if (C == 1) {
        hipLaunchKernelGGL(reward_events_partial_kernel, dim3(1), dim3(EVENT_BLOCK), 0, stream, rewards, E, A, partial,
                           totals, first_rows);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(reward_events_partial_kernel, dim3(C), dim3(EVENT_BLOCK), 0, stream, rewards, E, A, partial,
                       static_cast<uint64_t*>(nullptr), static_cast<double*>(nullptr));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(reward_events_finish_kernel, dim3(1), dim3(256), 0, stream, rewards, E * A, partial, C, totals,
                       first_rows);
    return hipGetLastError();
}

}  // namespace oc

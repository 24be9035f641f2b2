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

#include <mutex>
#include <string>

#include "../../include/marl_amd.h"

namespace oc {
int set_error(int code, const std::string& msg);  // csrc/capi.hip: records marl_last_error() of this thread

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

// ---------------------------------------------------------------------------------------
// Several steps at once (rewards [K][E * A], e.g. the rows a rollout has just stored): block (c, k)
// scans chunk c of step k, then one block adds everything to the totals and walks the steps in
// order for the rows-0..1 part, so the result equals K single-step calls bit for bit (the
// rows-0..1 return is a double accumulated step after step, like the reference's Python float).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(EVENT_BLOCK)
reward_events_many_partial_kernel(const float* __restrict__ rewards, int E, int A, uint64_t* __restrict__ partial) {
    const float* step_rewards = rewards + (size_t)blockIdx.y * (size_t)E * A;
    uint32_t tenths = 0, deliv = 0, plate = 0, pot = 0;
    for (int e = blockIdx.x * EVENT_BLOCK + threadIdx.x; e < E; e += gridDim.x * EVENT_BLOCK) {
        bool d = false, p = false, o = false;
        for (int a = 0; a < A; ++a) {
            const float r = step_rewards[(size_t)e * A + a];
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
        partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = s;
    }
}

__global__ void __launch_bounds__(256)
reward_events_many_finish_kernel(const float* __restrict__ rewards, int N, int K, const uint64_t* __restrict__ partial,
                                 int C, uint64_t* __restrict__ totals, double* __restrict__ first_rows) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint64_t s = 0;
    for (int c = lane; c < C * K; c += 64) s += partial[(size_t)c * 4 + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) totals[k] += s;
    if (threadIdx.x == 255 && first_rows)
        for (int step = 0; step < K; ++step) first_rows_update(rewards + (size_t)step * N, N, first_rows);
}

hipError_t launch_reward_events_many(const float* rewards, int K, int E, int A, uint64_t* partial, uint64_t* totals,
                                     double* first_rows, hipStream_t stream) {
    const int C = reward_event_chunks(E * A);
    hipLaunchKernelGGL(reward_events_many_partial_kernel, dim3(C, K), dim3(EVENT_BLOCK), 0, stream, rewards, E, A, partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(reward_events_many_finish_kernel, dim3(1), dim3(256), 0, stream, rewards, E * A, K, partial, C, totals,
                       first_rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// The same pass for callers that may not allocate (code under hipGraph capture, serving loops,
// harnesses with a no-allocation rule): the per-chunk partial sums, the one piece of scratch this
// pass needs, live in a file-scope device array instead of a caller-supplied workspace.  Every
// GPU the library runs on holds its own copy of the array (8 KiB) at its own address until the
// process ends; the host learns that address once per device and passes it to the kernels like
// any other buffer.
// ---------------------------------------------------------------------------------------
constexpr int EVENT_SCRATCH_CHUNKS = 256;  // reward_event_chunks() never returns more
static __device__ __align__(256) uint64_t scratch[EVENT_SCRATCH_CHUNKS * 4];

constexpr int MAX_DEVICES = 64;
static std::mutex g_scratch_mutex;
static uint64_t* g_scratch_ptr[MAX_DEVICES] = {nullptr};  // device address of `scratch`, per device

// device address of the current device's copy of `scratch` (looked up on the first call per device)
static int scratch_for_current_device(uint64_t** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(MARL_ERR_HIP, std::string("hipGetDevice: ") + hipGetErrorString(e));
    if (dev < 0 || dev >= MAX_DEVICES) return set_error(MARL_ERR_INVALID, "device index outside [0, 64)");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return set_error(MARL_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<uint64_t*>(p);
    }
    *out = g_scratch_ptr[dev];
    return MARL_OK;
}

}  // namespace oc

extern "C" int marl_reward_events_noalloc(const float* d_rewards, int num_envs, int num_agents, uint64_t* d_totals,
                                          double* d_first_rows, void* stream) {
    using namespace oc;
    if (!d_rewards || !d_totals) return set_error(MARL_ERR_INVALID, "NULL argument to marl_reward_events_noalloc");
    if (num_envs <= 0 || num_agents <= 0)
        return set_error(MARL_ERR_INVALID, "num_envs and num_agents must be positive");
    if ((int64_t)num_envs * num_agents > (int64_t)MARL_REWARD_EVENTS_NOALLOC_MAX_ROWS)
        return set_error(MARL_ERR_INVALID, "marl_reward_events_noalloc handles at most " +
                                               std::to_string(MARL_REWARD_EVENTS_NOALLOC_MAX_ROWS) +
                                               " rows per call (num_envs * num_agents); use marl_reward_events with a workspace");
    if (((uintptr_t)d_totals & 7u) != 0 || ((uintptr_t)d_first_rows & 7u) != 0 || ((uintptr_t)d_rewards & 3u) != 0)
        return set_error(MARL_ERR_INVALID, "d_totals and d_first_rows must be 8-byte aligned, d_rewards 4-byte aligned");
    if (reward_event_chunks(num_envs * num_agents) > EVENT_SCRATCH_CHUNKS)
        return set_error(MARL_ERR_INVALID, "row count needs more partial sums than the static scratch holds");
    uint64_t* partial = nullptr;
    const int rc = scratch_for_current_device(&partial);
    if (rc != MARL_OK) return rc;
    // every chunk sum the finishing pass reads is written by the first pass of the same call:
    // nothing in the scratch has to start at zero
    hipError_t e = launch_reward_events(d_rewards, num_envs, num_agents, partial, d_totals, d_first_rows,
                                        reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return set_error(MARL_ERR_HIP, std::string("reward event kernel launch: ") + hipGetErrorString(e));
    return MARL_OK;
}


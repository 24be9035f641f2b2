// GAE / value-target pass and per-step row sums for gfx950.
//
// Follows agents/MAPPO.py:82-114,144 and agents/CentralizedMAPPO.py:48-82 literally:
//   mask_t   = 1 - dones[t+1]          (for t = T-1: 1 - dones[T-1], the flag stored WITH
//                                        the last row, MAPPO.py:102-104)
//   delta_t  = r_t + gamma * V_{t+1} * mask_t - V_t
//   adv_t    = delta_t + (gamma*lam) * mask_t * adv_{t+1}
//   ret_t    = adv_t + V_t
// in float32 with one rounding per operation (no fused multiply-add), so that a float32
// tensor program and this kernel agree bit for bit; gamma*lam is formed in double first, as
// Python does.  HBM-bound: 12 bytes read + 8 bytes written per (t, row) sample; one lane
// owns one column and walks t downwards, adjacent lanes touch adjacent addresses.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oc {

// Measured on MI355X at T=128, N=131,072 (335 MB of traffic): blocks of 6 steps beat 4, 8 and 12
// (48.8 / 50.6 / 50.4 / 60.0 us), and storing the two outputs with the non-temporal hint takes the
// launch from 62 us to 49 us: the 134 MB written would otherwise push the three input planes the
// scan is still reading out of the last-level cache.
#ifndef OC_GAE_UNROLL
#define OC_GAE_UNROLL 6
#endif
constexpr int GAE_UNROLL = OC_GAE_UNROLL;
// Input planes larger than this (3 * T * N * 4 bytes) cannot have stayed in the 256 MiB last-level
// cache since the rollout wrote them (an update's worth of other traffic lies in between): they are
// then read with the non-temporal hint.  Measured at T=128, N=131,072 (201 MB of inputs): inputs
// that really are cold (four input sets rotated) 62.2 us plain / 56.5 us non-temporal; the same
// inputs every launch (they stay cached) 48.8 / 53.9 us.  Blocks of 3 or 12 steps lose in both cases.
constexpr size_t GAE_NT_INPUT_BYTES = (size_t)64 << 20;

template <bool NT_LOADS>
__global__ void __launch_bounds__(256)
gae_kernel(const float* __restrict__ rewards, const float* __restrict__ dones, const float* __restrict__ values,
           const float* __restrict__ next_values, int T, int N, float gamma, float gl,
           float* __restrict__ adv, float* __restrict__ ret, int mode, const float* __restrict__ rmean) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    auto GAE_LD = [](const float* p) { return NT_LOADS ? __builtin_nontemporal_load(p) : *p; };
    float last = 0.0f;
    float v_next = mode == 1 ? next_values[0] : next_values[n];
    float d_next = dones[(size_t)(T - 1) * N + n];
    int t = T - 1;
    // blocks of GAE_UNROLL steps: all loads of a block are issued before the dependent chain
    for (; t >= GAE_UNROLL - 1; t -= GAE_UNROLL) {
        float r[GAE_UNROLL], v[GAE_UNROLL], d[GAE_UNROLL];
#pragma unroll
        for (int k = 0; k < GAE_UNROLL; ++k) {
            const size_t idx = (size_t)(t - k) * N + n;
            r[k] = mode == 1 ? rmean[t - k] : GAE_LD(rewards + idx);
            v[k] = GAE_LD(values + idx);
            d[k] = GAE_LD(dones + idx);
        }
#pragma unroll
        for (int k = 0; k < GAE_UNROLL; ++k) {
            const size_t idx = (size_t)(t - k) * N + n;
            const float mask = __fsub_rn(1.0f, d_next);
            const float a1 = __fmul_rn(gamma, v_next);
            const float a2 = __fmul_rn(a1, mask);
            const float a3 = __fadd_rn(r[k], a2);
            const float delta = __fsub_rn(a3, v[k]);
            const float b1 = __fmul_rn(gl, mask);
            const float b2 = __fmul_rn(b1, last);
            last = __fadd_rn(delta, b2);
            __builtin_nontemporal_store(last, &adv[idx]);
            if (ret) __builtin_nontemporal_store(__fadd_rn(last, v[k]), &ret[idx]);
            v_next = v[k];
            d_next = d[k];
        }
    }
    for (; t >= 0; --t) {
        const size_t idx = (size_t)t * N + n;
        const float r = mode == 1 ? rmean[t] : rewards[idx];
        const float v = values[idx];
        const float d = dones[idx];
        const float mask = __fsub_rn(1.0f, d_next);
        const float a1 = __fmul_rn(gamma, v_next);
        const float a2 = __fmul_rn(a1, mask);
        const float a3 = __fadd_rn(r, a2);
        const float delta = __fsub_rn(a3, v);
        const float b1 = __fmul_rn(gl, mask);
        const float b2 = __fmul_rn(b1, last);
        last = __fadd_rn(delta, b2);
        __builtin_nontemporal_store(last, &adv[idx]);
        if (ret) __builtin_nontemporal_store(__fadd_rn(last, v), &ret[idx]);
        v_next = v;
        d_next = d;
    }
}

// Row sums in two fixed-order passes (reproducible run to run): pass 1, workgroup (c, t) adds
// chunk c of row t in double; pass 2, one lane per row adds that row's chunk sums in order.
__global__ void __launch_bounds__(256)
row_partial_kernel(const float* __restrict__ x, int N, int chunk, int C, double* __restrict__ partial) {
    const int c = blockIdx.x, t = blockIdx.y;
    const float* row = x + (size_t)t * N;
    const int lo = c * chunk;
    const int hi = min(N, lo + chunk);
    double acc = 0.0;
    for (int n = lo + (int)threadIdx.x; n < hi; n += (int)blockDim.x) acc += (double)row[n];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)t * C + c] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ void row_finish_kernel(const double* __restrict__ partial, int T, int C, int N, double* __restrict__ sums,
                                  float* __restrict__ mean_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += partial[(size_t)t * C + c];
    if (sums) sums[t] = s;
    if (mean_out) mean_out[t] = (float)(s / (double)N);
}

int row_sum_chunks(int N) {
    int c = (N + 8191) / 8192;
    return c < 1 ? 1 : c > 64 ? 64 : c;
}

hipError_t launch_gae(const float* rewards, const float* dones, const float* values, const float* next_values,
                      int T, int N, float gamma, float gl, float* adv, float* ret, int mode,
                      const float* rmean, hipStream_t stream) {
    const int block = 256;
    const int grid = (N + block - 1) / block;
    if ((size_t)3 * (size_t)T * (size_t)N * sizeof(float) > GAE_NT_INPUT_BYTES)
        hipLaunchKernelGGL(gae_kernel<true>, dim3(grid), dim3(block), 0, stream, rewards, dones, values, next_values, T, N,
                           gamma, gl, adv, ret, mode, rmean);
    else
        hipLaunchKernelGGL(gae_kernel<false>, dim3(grid), dim3(block), 0, stream, rewards, dones, values, next_values, T, N,
                           gamma, gl, adv, ret, mode, rmean);
    return hipGetLastError();
}

hipError_t launch_row_sums(const float* x, int T, int N, double* partial, double* sums, float* mean_out,
                           hipStream_t stream) {
    const int C = row_sum_chunks(N);
    const int chunk = (N + C - 1) / C;
    hipLaunchKernelGGL(row_partial_kernel, dim3(C, T), dim3(256), 0, stream, x, N, chunk, C, partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(row_finish_kernel, dim3((T + 63) / 64), dim3(64), 0, stream, partial, T, C, N, sums, mean_out);
    return hipGetLastError();
}

}  // namespace oc

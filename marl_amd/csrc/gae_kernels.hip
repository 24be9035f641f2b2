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

constexpr int GAE_UNROLL = 8;

__global__ void __launch_bounds__(256)
gae_kernel(const float* __restrict__ rewards, const float* __restrict__ dones, const float* __restrict__ values,
           const float* __restrict__ next_values, int T, int N, float gamma, float gl,
           float* __restrict__ adv, float* __restrict__ ret, int mode, const float* __restrict__ rmean) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
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
            r[k] = mode == 1 ? rmean[t - k] : rewards[idx];
            v[k] = values[idx];
            d[k] = dones[idx];
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
            adv[idx] = last;
            if (ret) ret[idx] = __fadd_rn(last, v[k]);
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
        adv[idx] = last;
        if (ret) ret[idx] = __fadd_rn(last, v);
        v_next = v;
        d_next = d;
    }
}

// one workgroup per row t: sum of x[t][0..N) in double, fixed order => reproducible
__global__ void __launch_bounds__(256)
row_sum_kernel(const float* __restrict__ x, int N, double* __restrict__ sums, float* __restrict__ mean_out) {
    const int t = blockIdx.x;
    const float* row = x + (size_t)t * N;
    double acc = 0.0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) acc += (double)row[n];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += part[w];
        if (sums) sums[t] = s;
        if (mean_out) mean_out[t] = (float)(s / (double)N);
    }
}

hipError_t launch_gae(const float* rewards, const float* dones, const float* values, const float* next_values,
                      int T, int N, float gamma, float gl, float* adv, float* ret, int mode,
                      const float* rmean, hipStream_t stream) {
    const int block = 256;
    const int grid = (N + block - 1) / block;
    hipLaunchKernelGGL(gae_kernel, dim3(grid), dim3(block), 0, stream, rewards, dones, values, next_values, T, N,
                       gamma, gl, adv, ret, mode, rmean);
    return hipGetLastError();
}

hipError_t launch_row_sums(const float* x, int T, int N, double* sums, float* mean_out, hipStream_t stream) {
    hipLaunchKernelGGL(row_sum_kernel, dim3(T), dim3(256), 0, stream, x, N, sums, mean_out);
    return hipGetLastError();
}

}  // namespace oc

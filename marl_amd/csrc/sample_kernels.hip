// Categorical action sampling of the rollout, model.py:52-56 (`Categorical(logits=...)`,
// `.sample()`, `.log_prob(action)`, `.entropy()`), as one launch instead of about fifteen small
// tensor ops.  One lane per row:
//   m = max_k l_k;  e_k = exp(l_k - m);  S = sum_k e_k (k ascending);  lse = m + log S
//   action   = first k with e_0 + ... + e_k > u * S        (inverse CDF, u uniform in [0, 1))
//   log_prob = l_action - lse
//   entropy  = - sum_k (e_k / S) * (l_k - lse)
// u comes from a counter RNG keyed by (seed, call counter, global row), so a rollout is
// reproducible and a shard of a larger job draws what the whole job would have drawn.
// Rows may be wider than K (row stride `ld`): a caller that computes the critic's value in the
// same GEMM as the logits (one more output column) gets that column copied out alongside.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oc {

__device__ __forceinline__ uint32_t smix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256)
sample_categorical_kernel(const float* __restrict__ logits, int N, int K, int ld, uint32_t key,
                          const uint64_t* __restrict__ counter, uint64_t row_base, long long* __restrict__ actions,
                          float* __restrict__ logprob, float* __restrict__ entropy, float* __restrict__ next_col) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* l = logits + (size_t)n * ld;
    float m = l[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, l[k]);
    float S = 0.0f;
    for (int k = 0; k < K; ++k) S = __fadd_rn(S, expf(__fsub_rn(l[k], m)));
    const float lse = __fadd_rn(m, logf(S));

    const uint64_t c = counter ? counter[0] : 0;
    const uint64_t row = row_base + (uint64_t)n;
    uint32_t r = smix32(key ^ (uint32_t)c);
    r = smix32(r ^ (uint32_t)(c >> 32) ^ 0x85EBCA6Bu);
    r = smix32(r ^ (uint32_t)row);
    r = smix32(r + (uint32_t)(row >> 32));
    const float u = (float)(r >> 8) * 5.9604644775390625e-08f;  // 24 bits -> [0, 1)
    const float target = __fmul_rn(u, S);

    float cum = 0.0f, ent = 0.0f;
    int a = K - 1;
    bool found = false;
    for (int k = 0; k < K; ++k) {
        const float e = expf(__fsub_rn(l[k], m));
        cum = __fadd_rn(cum, e);
        if (!found && cum > target) { a = k; found = true; }
        ent = __fsub_rn(ent, __fmul_rn(__fdiv_rn(e, S), __fsub_rn(l[k], lse)));
    }
    actions[n] = a;
    if (logprob) logprob[n] = __fsub_rn(l[a], lse);
    if (entropy) entropy[n] = ent;
    if (next_col) next_col[n] = l[K];
}

hipError_t launch_sample_categorical(const float* logits, int N, int K, int ld, uint32_t key, const uint64_t* counter,
                                     uint64_t row_base, long long* actions, float* logprob, float* entropy,
                                     float* next_col, hipStream_t stream) {
    hipLaunchKernelGGL(sample_categorical_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, logits, N, K, ld, key, counter,
                       row_base, actions, logprob, entropy, next_col);
    return hipGetLastError();
}

}  // namespace oc

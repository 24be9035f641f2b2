// The clipped-surrogate loss of one PPO minibatch (agents/MAPPO.py:167-200, model.py:52-56) and
// its gradient with respect to the two network heads, as two launches instead of the ~70 small
// tensor ops autograd runs for the same arithmetic.  One lane per sample:
//   lp_k   = l_k - logsumexp(l)                    (Categorical(logits=l))
//   newlp  = lp_a,  H = -sum_k p_k lp_k,  logratio = newlp - oldlp,  r = exp(logratio)
//   pg     = max(-A r, -A clamp(r, 1-c, 1+c))
//   vl     = 0.5 (v - target)^2
//   loss   = mean(pg) - entropy_coef * mean(H) + value_coef * mean(vl)
// and, the loss being a plain mean over samples, every sample's gradient is known on the spot:
//   d loss / d l_k = [g_r r (1[k = a] - p_k) + entropy_coef p_k (lp_k + H)] / M
//   d loss / d v   = value_coef (v - target) / M
// with g_r the slope of pg in r (-A on the unclipped branch, 0 where the clipped branch wins and
// r lies outside the clip range; at a tie autograd gives each branch half, which adds up to -A
// inside the range).  The six logged statistics are reduced in double.
// Samples may be addressed through an index into the rollout store (the minibatch permutation),
// which also saves the four gather launches in front of the loss.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oc {

constexpr int PPO_BLOCK = 256;
constexpr int PPO_STATS = 6;  // value loss, policy loss, entropy, approx_kl, old_approx_kl, clip fraction (sums)

int ppo_loss_chunks(int M) {
    int c = (M + PPO_BLOCK * 4 - 1) / (PPO_BLOCK * 4);
    return c < 1 ? 1 : c > 512 ? 512 : c;
}

template <int K>
__global__ void __launch_bounds__(PPO_BLOCK)
ppo_loss_partial_kernel(const float* __restrict__ logits, int M, int ld, const float* __restrict__ values,
                        const long long* __restrict__ index, const float* __restrict__ actions_f,
                        const float* __restrict__ old_logprob, const float* __restrict__ advantages,
                        const float* __restrict__ value_target, float clip, float entropy_coef, float value_coef,
                        float* __restrict__ grad_logits, float* __restrict__ grad_values, double* __restrict__ partial) {
    double acc[PPO_STATS] = {0, 0, 0, 0, 0, 0};
    const float inv_m = 1.0f / (float)M;
    for (int n = blockIdx.x * PPO_BLOCK + threadIdx.x; n < M; n += gridDim.x * PPO_BLOCK) {
        const size_t src = index ? (size_t)index[n] : (size_t)n;
        const float* l = logits + (size_t)n * ld;
        float lk[K];
        float m = l[0];
#pragma unroll
        for (int k = 0; k < K; ++k) { lk[k] = l[k]; m = fmaxf(m, lk[k]); }
        float S = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) S += expf(lk[k] - m);
        const float lse = m + logf(S);
        int a = (int)actions_f[src];
        a = a < 0 ? 0 : a >= K ? K - 1 : a;
        float H = 0.0f, lp_a = 0.0f;
        float p[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float lp = lk[k] - lse;
            p[k] = expf(lp);
            H -= p[k] * lp;
            lp_a = (k == a) ? lp : lp_a;
        }
        const float logratio = lp_a - old_logprob[src];
        const float r = expf(logratio);
        const float A = advantages[src];
        const float lo = 1.0f - clip, hi = 1.0f + clip;
        const float rc = fminf(fmaxf(r, lo), hi);
        const float t1 = -A * r, t2 = -A * rc;
        const bool inside = r >= lo && r <= hi;
        const float pg = fmaxf(t1, t2);
        const float g_r = t1 > t2 ? -A : t1 == t2 ? (inside ? -A : -0.5f * A) : (inside ? -A : 0.0f);
        const float v = values[n], dv = v - value_target[src];

        const float w = g_r * r * inv_m, we = entropy_coef * inv_m;
        float* gl = grad_logits + (size_t)n * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float lp = lk[k] - lse;
            gl[k] = w * ((k == a ? 1.0f : 0.0f) - p[k]) + we * p[k] * (lp + H);
        }
        grad_values[n] = value_coef * dv * inv_m;

        acc[0] += 0.5 * (double)dv * (double)dv;
        acc[1] += (double)pg;
        acc[2] += (double)H;
        acc[3] += (double)((r - 1.0f) - logratio);
        acc[4] += (double)(-logratio);
        acc[5] += fabsf(r - 1.0f) > clip ? 1.0 : 0.0;
    }
    __shared__ double part[PPO_BLOCK / 64][PPO_STATS];
#pragma unroll
    for (int k = 0; k < PPO_STATS; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[k] += __shfl_down(acc[k], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < PPO_STATS; ++k) part[threadIdx.x >> 6][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < PPO_STATS) {
        double s = 0;
        for (int w = 0; w < PPO_BLOCK / 64; ++w) s += part[w][threadIdx.x];
        partial[(size_t)blockIdx.x * PPO_STATS + threadIdx.x] = s;
    }
}

// stats[0] = loss; stats[1..6] = means of the six statistics, in the order the learner logs them.
// One wave per statistic: lanes stride over the chunk sums, then a shuffle reduction.
__global__ void __launch_bounds__(64 * PPO_STATS)
ppo_loss_finish_kernel(const double* __restrict__ partial, int C, int M, float entropy_coef, float value_coef,
                       double* __restrict__ stats) {
    __shared__ double mean[PPO_STATS];
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double s = 0;
    for (int c = lane; c < C; c += 64) s += partial[(size_t)c * PPO_STATS + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
        mean[k] = s / (double)M;
        stats[1 + k] = mean[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) stats[0] = mean[1] - (double)entropy_coef * mean[2] + (double)value_coef * mean[0];
}

hipError_t launch_ppo_loss(const float* logits, int M, int K, int ld, const float* values, const long long* index,
                           const float* actions_f, const float* old_logprob, const float* advantages,
                           const float* value_target, float clip, float entropy_coef, float value_coef,
                           float* grad_logits, float* grad_values, double* stats, double* partial, hipStream_t stream) {
    const int C = ppo_loss_chunks(M);
#define OC_PPO_CASE(KK)                                                                                                  \
    case KK:                                                                                                             \
        hipLaunchKernelGGL(ppo_loss_partial_kernel<KK>, dim3(C), dim3(PPO_BLOCK), 0, stream, logits, M, ld, values, index, \
                           actions_f, old_logprob, advantages, value_target, clip, entropy_coef, value_coef, grad_logits,  \
                           grad_values, partial);                                                                        \
        break;
    switch (K) {
        OC_PPO_CASE(2) OC_PPO_CASE(3) OC_PPO_CASE(4) OC_PPO_CASE(5) OC_PPO_CASE(6) OC_PPO_CASE(7) OC_PPO_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef OC_PPO_CASE
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ppo_loss_finish_kernel, dim3(1), dim3(64 * PPO_STATS), 0, stream, partial, C, M, entropy_coef, value_coef, stats);
    return hipGetLastError();
}

}  // namespace oc

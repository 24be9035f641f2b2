// Second half of the split form of env.step: int8 rows -> float32 observation rows.
//
// Observation tensors larger than the last-level cache have to drain to HBM while they are
// written.  How fast that goes depends on how narrow the front of addresses is that the chip
// writes at any moment: a wave of the fused kernel owns 64 rows (51.7 KB at D = 202) and all waves
// of a launch stream at once, i.e. thousands of 1 KiB-granular streams spread over the whole tensor
// (about 5.4 TB/s); workgroups that write 4 KiB and end, dispatched in address order, keep the front
// a few MB wide (6.8 TB/s for a plain fill).  So the simulation kernel leaves the rows as int8 (a
// quarter of the bytes, cache resident) and this kernel, one 16-byte store per lane and workgroup
// life, widens them.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// U dwords of int8 per lane, loaded up front (each load instruction of a wave covers 256
// consecutive bytes), then U 16-byte stores (each covering 1 KiB): a workgroup of BS lanes writes
// BS * U * 16 consecutive bytes and ends.
template <int NT, int U, int BS>
__global__ void __launch_bounds__(BS)
expand_kernel(const uint32_t* __restrict__ img, float* __restrict__ out, uint32_t nwords, uint32_t ntail) {
    const uint32_t g0 = blockIdx.x * (uint32_t)(BS * U) + threadIdx.x;
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t g = g0 + (uint32_t)(u * BS);
        v[u] = g < nwords ? __builtin_nontemporal_load(img + g) : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t g = g0 + (uint32_t)(u * BS);
        if (g < nwords) {
            f32x4 f;
            f.x = (float)(int)(int8_t)(v[u]);
            f.y = (float)(int)(int8_t)(v[u] >> 8);
            f.z = (float)(int)(int8_t)(v[u] >> 16);
            f.w = (float)(int)(int8_t)(v[u] >> 24);
            f32x4* dst = reinterpret_cast<f32x4*>(out) + g;
            if constexpr (NT == 1) __builtin_nontemporal_store(f, dst);
            else *dst = f;
        } else if (g - nwords < ntail) {
            // up to 3 values past the last whole group of four
            const size_t idx = (size_t)nwords * 4 + (g - nwords);
            out[idx] = (float)(int)reinterpret_cast<const int8_t*>(img)[idx];
        }
    }
}

template <int NT, int U, int BS>
static hipError_t launch_expand_as(const void* img, float* out, size_t count, hipStream_t stream) {
    const uint32_t nwords = (uint32_t)(count / 4), ntail = (uint32_t)(count % 4);
    const size_t per_block = (size_t)BS * U;
    const unsigned grid = (unsigned)(((size_t)nwords + ntail + per_block - 1) / per_block);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL((expand_kernel<NT, U, BS>), dim3(grid), dim3(BS), 0, stream, reinterpret_cast<const uint32_t*>(img), out,
                       nwords, ntail);
    return hipGetLastError();
}

// img: int8 [count]; out: float32 [count], 16-byte aligned.
// policy: bit 0 non-temporal stores; bits 1-2 log2(dwords per lane); bit 3 one-wave workgroups
hipError_t launch_expand(const void* img, float* out, size_t count, int policy, hipStream_t stream) {
    if (count / 4 + 3 > 0xFFFFFF00u) return hipErrorInvalidValue;
    switch (policy & 15) {
#define OC_CASE(P, NT, U, BS) case P: return launch_expand_as<NT, U, BS>(img, out, count, stream)
        OC_CASE(0, 0, 1, 256); OC_CASE(1, 1, 1, 256); OC_CASE(2, 0, 2, 256); OC_CASE(3, 1, 2, 256);
        OC_CASE(4, 0, 4, 256); OC_CASE(5, 1, 4, 256); OC_CASE(6, 0, 8, 256); OC_CASE(7, 1, 8, 256);
        OC_CASE(8, 0, 1, 64); OC_CASE(9, 1, 1, 64); OC_CASE(10, 0, 2, 64); OC_CASE(11, 1, 2, 64);
        OC_CASE(12, 0, 4, 64); OC_CASE(13, 1, 4, 64); OC_CASE(14, 0, 8, 64); OC_CASE(15, 1, 8, 64);
#undef OC_CASE
    }
    return hipErrorInvalidValue;
}

}  // namespace oc

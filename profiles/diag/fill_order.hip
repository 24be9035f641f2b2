// A wave streams its own 51,712-byte region (51 stores of 1 KiB, sc1) beyond the last-level cache:
// does the ORDER in which it issues them matter?  Consecutive KiB of a region share a 4 KiB block
// of addresses; if the L2 / fabric spreads addresses over its channels in blocks of that size, four
// consecutive stores of a wave queue at one channel.
//   ORDER 0  k = 0, 1, 2, ... (what the env kernel does)
//   ORDER s  stride s KiB: 0, s, 2s, ..., then 1, 1+s, ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int S, int K> struct Perm {  // K-th store in issue order -> KiB index (52 slots, slot 51 unused)
    static constexpr int per = S == 0 ? 1 : (52 + (S == 0 ? 1 : S) - 1) / (S == 0 ? 1 : S);
    static constexpr int value = S == 0 ? K : (K % per) * S + K / per;
};

template <int S, int K>
__device__ __forceinline__ void stores(const __amdgpu_buffer_rsrc_t& rsrc, uint32_t voff, i32x4 v) {
    if constexpr (K < (S == 0 ? 51 : Perm<S, 0>::per * S)) {
        constexpr int kib = Perm<S, K>::value;
        if constexpr (kib < 51) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff + kib * 1024, 0, 16);
        stores<S, K + 1>(rsrc, voff, v);
    }
}

template <int S>
__global__ void __launch_bounds__(256) fill_regions(float* out, uint32_t nwaves) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const uint64_t bits = reinterpret_cast<uint64_t>(reinterpret_cast<char*>(out) + wave * 51712);
    char* ubase = reinterpret_cast<char*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, 51712, 0x00020000);
    i32x4 v = {1, 2, 3, 4};
    stores<S, 0>(rsrc, (uint32_t)lane * 16, v);
}

// lanes of one store spread over four 4 KiB blocks: lane group g (16 lanes, 256 B) writes into block 4 * j + g
__global__ void __launch_bounds__(256) fill_regions_spread(float* out, uint32_t nwaves) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const uint64_t bits = reinterpret_cast<uint64_t>(reinterpret_cast<char*>(out) + wave * 51712);
    char* ubase = reinterpret_cast<char*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, 51712, 0x00020000);
    i32x4 v = {1, 2, 3, 4};
    const uint32_t g = lane >> 4, l = lane & 15;
    // 16 KiB super-blocks: within one, store q (0..15) has lane group g write 256 B at block g, offset q * 256
#pragma unroll
    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
        for (int q = 0; q < 16; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (uint32_t)(sb * 16384) + g * 4096u + (uint32_t)(q * 256) + l * 16u, 0, 16);
}

template <class F>
static double run(F launch, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 4; ++i) launch(i);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch(i);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / iters;
}

int main() {
    const size_t nbytes = (size_t)524288 * 2 * 202 * 4;
    float *b0, *b1;
    if (hipMalloc(&b0, nbytes) != hipSuccess || hipMalloc(&b1, nbytes) != hipSuccess) return 1;
    const unsigned nw = (unsigned)(nbytes / 51712), gr = (nw + 3) / 4;
#define ROW(S) printf("  stride %2d KiB: %.2f TB/s\n", S, nbytes / 1e6 / run([&](int i) { hipLaunchKernelGGL(fill_regions<S>, dim3(gr), dim3(256), 0, 0, i & 1 ? b1 : b0, nw); }, 40));
    printf("847 MB per launch, own 51,712-byte region per wave, issue order of the 51 stores:\n");
    ROW(0) ROW(2) ROW(4) ROW(8) ROW(13) ROW(16) ROW(26)
    printf("  lanes of a store spread over four 4 KiB blocks: %.2f TB/s\n",
           nbytes / 1e6 / run([&](int i) { hipLaunchKernelGGL(fill_regions_spread, dim3(gr), dim3(256), 0, 0, i & 1 ? b1 : b0, nw); }, 40));
    return 0;
}

// The env kernel's store stream (a 51 KiB region per wave, 51 stores of 1 KiB, 2048 waves, ten
// targets in turn) with the regions dealt to the waves in another order: wave w writes region
// (w * mult) mod 2048 (mult odd: a permutation), or a pseudo-random permutation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <random>

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 51;

__global__ void __launch_bounds__(256) fill(char* out, const uint32_t* perm) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t r = __builtin_amdgcn_readfirstlane(perm[w]);
    char* base = out + (size_t)r * K * 1024;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(((uint64_t)hi << 32) | lo), 0, K * 1024, 0x00020000);
    i32x4 v = {(int)w, 2, 3, 4};
#pragma unroll
    for (int k = 0; k < K; ++k) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane * 16 + k * 1024, 0, 16);
}

template <class F>
static double run(F launch, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 12; ++i) launch(i);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch(i);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / iters;
}

int main() {
    const uint32_t W = 2048;
    const size_t nbytes = (size_t)W * K * 1024;
    constexpr int NB = 10;
    char* buf[NB];
    for (auto& b : buf) if (hipMalloc((void**)&b, nbytes) != hipSuccess) return 1;
    uint32_t* dperm;
    if (hipMalloc((void**)&dperm, W * 4) != hipSuccess) return 1;
    std::vector<uint32_t> perm(W);
    auto go = [&](const char* name) {
        (void)hipMemcpy(dperm, perm.data(), W * 4, hipMemcpyHostToDevice);
        const double t = run([&](int i) { hipLaunchKernelGGL(fill, dim3(W / 4), dim3(256), 0, 0, buf[i % NB], dperm); }, 60);
        printf("%-28s %.2f us (%.2f TB/s)\n", name, t, nbytes / 1e6 / t);
    };
    for (uint32_t mult : {1u, 3u, 5u, 17u, 33u, 257u, 1001u}) {
        for (uint32_t w = 0; w < W; ++w) perm[w] = (w * mult) % W;
        char name[64]; snprintf(name, sizeof name, "region (w * %u) mod 2048", mult);
        go(name);
    }
    std::mt19937 rng(1);
    for (int trial = 0; trial < 3; ++trial) {
        for (uint32_t w = 0; w < W; ++w) perm[w] = w;
        std::shuffle(perm.begin(), perm.end(), rng);
        go("random permutation");
    }
    // the four waves of a workgroup keep neighbouring regions, workgroups shuffled
    for (uint32_t g = 0; g < W / 4; ++g) perm[g * 4] = g;
    std::vector<uint32_t> groups(W / 4);
    for (uint32_t g = 0; g < W / 4; ++g) groups[g] = g;
    std::shuffle(groups.begin(), groups.end(), rng);
    for (uint32_t g = 0; g < W / 4; ++g) for (uint32_t j = 0; j < 4; ++j) perm[g * 4 + j] = groups[g] * 4 + j;
    go("workgroups shuffled");
    return 0;
}

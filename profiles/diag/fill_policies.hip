// Store cache policy x store pattern beyond the last-level cache (847 MB, two buffers alternating,
// 40 launches back to back): 256-lane workgroups with one 16-byte store per lane in address order,
// against waves that stream their own 51,712-byte region.  aux bits: 0 plain, 2 nt, 16 sc1, 18 sc1+nt, 1 sc0, 17 sc0+sc1.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ void __launch_bounds__(256) fill_short(float* out, uint32_t nbytes_lo, uint32_t chunk_base) {
    // one descriptor per workgroup over its own 4 KiB
    char* base = reinterpret_cast<char*>(out) + ((size_t)blockIdx.x + chunk_base) * 4096;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 4096, 0x00020000);
    i32x4 v = {1, 2, 3, 4};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, threadIdx.x * 16, 0, AUX);
}

template <int AUX>
__global__ void __launch_bounds__(256) fill_regions(float* out, uint32_t nwaves) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nwaves) return;
    char* base = reinterpret_cast<char*>(out) + wave * 51712;
    const uint64_t bits = reinterpret_cast<uint64_t>(base);
    char* ubase = reinterpret_cast<char*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, 51712, 0x00020000);
    i32x4 v = {1, 2, 3, 4};
#pragma unroll
    for (int k = 0; k < 51; ++k) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (uint32_t)(k * 1024 + lane * 16), 0, AUX);
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
    for (size_t envs : {65536ul, 524288ul}) {
        const size_t nbytes = envs * 2 * 202 * 4;  // a multiple of 4096 and of 51712
        float *b0, *b1;
        if (hipMalloc(&b0, nbytes) != hipSuccess || hipMalloc(&b1, nbytes) != hipSuccess) return 1;
        const unsigned gs = (unsigned)(nbytes / 4096), nw = (unsigned)(nbytes / 51712), gr = (nw + 3) / 4;
        printf("%zu MB per launch, TB/s by aux policy:\n", nbytes / 1000000);
#define ROW(AUX)                                                                                                           \
    {                                                                                                                      \
        const double a = run([&](int i) { hipLaunchKernelGGL(fill_short<AUX>, dim3(gs), dim3(256), 0, 0, i & 1 ? b1 : b0, 0u, 0u); }, 40); \
        const double b = run([&](int i) { hipLaunchKernelGGL(fill_regions<AUX>, dim3(gr), dim3(256), 0, 0, i & 1 ? b1 : b0, nw); }, 40);    \
        printf("  aux %2d: one store per wave %.2f   own regions %.2f\n", AUX, nbytes / 1e6 / a, nbytes / 1e6 / b);             \
    }
        ROW(0) ROW(2) ROW(16) ROW(18) ROW(1) ROW(17)
        hipFree(b0); hipFree(b1);
    }
    return 0;
}

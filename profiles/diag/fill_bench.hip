// Floor for the headline launch: a kernel that only writes the 105.9 MB of observation rows with
// the geometry of env_kernel's store stream (2,048 waves, each streaming its 64 rows x 202 floats
// as 16-byte stores, 1 KiB per wave instruction) and nothing else.
// Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_bench profiles/diag/fill_bench.hip && /tmp/fill_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ void __launch_bounds__(256) fill(float* out, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t row0 = wave * 64;
    if (row0 >= (size_t)rows) return;
    const uint64_t bits = reinterpret_cast<uint64_t>(out + row0 * D);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32));
    float* base = reinterpret_cast<float*>(((uint64_t)hi << 32) | lo);
    const uint32_t nbytes = 64u * (uint32_t)D * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
    f32x4 v = {(float)lane, 1.0f, 2.0f, 3.0f};
#pragma unroll
    for (int k = 0; k < 51; ++k)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rsrc, (uint32_t)lane * 16u + (uint32_t)k * 1024u, 0, AUX);
}

template <int AUX>
static float run(float* buf0, float* buf1, int rows, int D, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (rows / 64 + 3) / 4;
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(fill<AUX>, dim3(grid), dim3(256), 0, 0, i & 1 ? buf1 : buf0, rows, D);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(fill<AUX>, dim3(grid), dim3(256), 0, 0, i & 1 ? buf1 : buf0, rows, D);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / iters;
}

int main() {
    const int rows = 131072, D = 202;
    float *b0, *b1;
    if (hipMalloc(&b0, (size_t)rows * D * 4) != hipSuccess || hipMalloc(&b1, (size_t)rows * D * 4) != hipSuccess) return 1;
    const double mb = (double)rows * D * 4 / 1e6;
    const float plain = run<0>(b0, b1, rows, D, 2000), nt = run<2>(b0, b1, rows, D, 2000), sc1 = run<16>(b0, b1, rows, D, 2000);
    printf("%.1f MB per launch, two buffers alternating: plain %.2f us (%.2f TB/s), nt %.2f us, sc1 %.2f us (%.2f TB/s)\n", mb,
           plain, mb / plain, nt, sc1, mb / sc1);
    return 0;
}

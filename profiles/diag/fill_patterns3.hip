// Third look at store patterns beyond the last-level cache: ONE-WAVE workgroups that write a region
// of R bytes front to back and end (what the env kernel's wave would do if it owned 64, 32, 16 or 8
// rows), with and without a stretch of dependent ALU work before the first store (the simulation).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_patterns3 profiles/diag/fill_patterns3.hip && /tmp/fill_patterns3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// region bytes = CHUNKS * 1024 + TAILB (TAILB < 1024, multiple of 16); ALU = dependent multiply-adds up front
template <int CHUNKS, int TAILB, int ALU, int AUX>
__global__ void __launch_bounds__(64) wave_region(float* out, size_t nbytes, float seed) {
    const int lane = threadIdx.x;
    constexpr size_t R = (size_t)CHUNKS * 1024 + TAILB;
    const size_t base = (size_t)blockIdx.x * R;
    float x = seed + lane;
#pragma unroll 1
    for (int i = 0; i < ALU; ++i) x = x * 1.0001f + 0.5f;  // ~8 cycles per iteration with one wave per SIMD
    f32x4 v = {x, 2.0f, 3.0f, 4.0f};
    const uint32_t lim = (uint32_t)(base + R <= nbytes ? R : (base < nbytes ? nbytes - base : 0));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out) + base, 0, lim, 0x00020000);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int k = 0; k < CHUNKS + (TAILB ? 1 : 0); ++k)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rsrc, (uint32_t)(k * 1024 + lane * 16), 0, AUX);
}

template <class K>
static double run(K kernel, float* b0, float* b1, size_t nbytes, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, i & 1 ? b1 : b0, nbytes, 1.0f);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, i & 1 ? b1 : b0, nbytes, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)nbytes / 1e6 / (ms * 1000.0 / iters);  // TB/s
}

#define ROW(NAME, CH, TB)                                                                                          \
    {                                                                                                              \
        const size_t R = (size_t)CH * 1024 + TB;                                                                   \
        const int g = (int)((nbytes + R - 1) / R);                                                                 \
        printf("%4zu MB  %-28s sc1: no ALU %.2f  ALU 400 %.2f  ALU 1200 %.2f | plain: no ALU %.2f  ALU 400 %.2f TB/s\n", \
               nbytes / 1000000, NAME, run(wave_region<CH, TB, 0, 16>, b0, b1, nbytes, g, it),                    \
               run(wave_region<CH, TB, 400, 16>, b0, b1, nbytes, g, it), run(wave_region<CH, TB, 1200, 16>, b0, b1, nbytes, g, it), \
               run(wave_region<CH, TB, 0, 0>, b0, b1, nbytes, g, it), run(wave_region<CH, TB, 400, 0>, b0, b1, nbytes, g, it)); \
    }

int main() {
    for (size_t envs : {65536ul, 524288ul}) {
        const size_t nbytes = envs * 2 * 202 * 4;
        float *b0, *b1;
        if (hipMalloc(&b0, nbytes) != hipSuccess || hipMalloc(&b1, nbytes) != hipSuccess) return 1;
        const int it = 40;
        ROW("64 rows, 51,712 B per wave", 50, 512)
        ROW("32 rows, 25,856 B per wave", 25, 256)
        ROW("16 rows, 12,928 B per wave", 12, 640)
        ROW(" 8 rows,  6,464 B per wave", 6, 320)
        ROW(" 4 rows,  3,232 B per wave", 3, 160)
        hipFree(b0); hipFree(b1);
    }
    return 0;
}

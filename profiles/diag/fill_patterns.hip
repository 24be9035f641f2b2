// Which store pattern reaches the HBM write rate of a plain device fill (6.8-6.9 TB/s) when the
// observation rows no longer fit the last-level cache?  Every variant writes rows x 202 floats
// with 16-byte buffer stores; they differ in which wave writes which KiB when.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_patterns profiles/diag/fill_patterns.hip && /tmp/fill_patterns
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 202, CH = 51;  // 64 rows x 202 floats = 51.7 KB = 50.5 KiB chunks per wave

template <int AUX>
__device__ __forceinline__ void store(float* base, uint32_t nbytes, uint32_t off) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
    f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rsrc, off, 0, AUX);
}

__device__ __forceinline__ float* uniform(float* p) {
    const uint64_t b = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b));
}

// P0: every wave streams its own 64 rows front to back (what env_kernel does)
template <int AUX>
__global__ void __launch_bounds__(256) p0(float* out, size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave * 64 >= rows) return;
    float* base = uniform(out + wave * 64 * D);
#pragma unroll
    for (int k = 0; k < CH; ++k) store<AUX>(base, 64u * D * 4u, lane * 16u + k * 1024u);
}

// P1: the four waves of a workgroup share the workgroup's 256 rows, KiB chunks dealt round robin
template <int AUX>
__global__ void __launch_bounds__(256) p1(float* out, size_t rows) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if ((size_t)blockIdx.x * 256 >= rows) return;
    float* base = uniform(out + (size_t)blockIdx.x * 256 * D);
#pragma unroll
    for (int k = 0; k < CH; ++k) store<AUX>(base, 256u * D * 4u, lane * 16u + (k * 4 + w) * 1024u);
    if (w < 2) store<AUX>(base, 256u * D * 4u, lane * 16u + (CH * 4 + w) * 1024u);  // 202 KiB = 4 * 50.5
}

// P2: like P0 but a lane writes 64 contiguous bytes (four stores), a wave instruction spans 4 KiB
template <int AUX>
__global__ void __launch_bounds__(256) p2(float* out, size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave * 64 >= rows) return;
    float* base = uniform(out + wave * 64 * D);
#pragma unroll
    for (int k = 0; k < CH; ++k) store<AUX>(base, 64u * D * 4u, lane * 64u + (k & 3) * 16u + (k >> 2) * 4096u);
}

// P3: what a plain device fill does: a workgroup writes 16 KiB and ends (many short workgroups)
template <int AUX>
__global__ void __launch_bounds__(256) p3(float* out, size_t rows) {
    const size_t nbytes_total = rows * D * 4;
    const size_t b0 = (size_t)blockIdx.x * 16384;
    if (b0 >= nbytes_total) return;
    float* base = uniform(reinterpret_cast<float*>(reinterpret_cast<char*>(out) + b0));
    const uint32_t nb = (uint32_t)(nbytes_total - b0 < 16384 ? nbytes_total - b0 : 16384);
#pragma unroll
    for (int j = 0; j < 4; ++j) store<AUX>(base, nb, (j * 256 + threadIdx.x) * 16u);
}

// P4: P0 with the occupancy cut by a dynamic LDS allocation
template <int AUX>
__global__ void __launch_bounds__(256) p4(float* out, size_t rows) {
    extern __shared__ float pad[];
    if (rows == 1) pad[threadIdx.x] = 0;  // keep the allocation
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave * 64 >= rows) return;
    float* base = uniform(out + wave * 64 * D);
#pragma unroll
    for (int k = 0; k < CH; ++k) store<AUX>(base, 64u * D * 4u, lane * 16u + k * 1024u);
}

template <class K>
static float run(K kernel, float* b0, float* b1, size_t rows, int grid, int iters, size_t lds = 0) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, i & 1 ? b1 : b0, rows);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, i & 1 ? b1 : b0, rows);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / iters;
}

int main() {
    for (size_t envs : {65536ul, 131072ul, 262144ul, 1048576ul}) {
        const size_t rows = envs * 2;
        float *b0, *b1;
        if (hipMalloc(&b0, rows * D * 4) != hipSuccess || hipMalloc(&b1, rows * D * 4) != hipSuccess) return 1;
        const double mb = (double)rows * D * 4 / 1e6;
        const int grid = (int)(rows / 256), it = envs > 300000 ? 40 : 200;
        const float a = run(p0<16>, b0, b1, rows, grid, it), b = run(p0<18>, b0, b1, rows, grid, it);
        const float c = run(p1<16>, b0, b1, rows, grid, it), d = run(p1<18>, b0, b1, rows, grid, it);
        const float e = run(p2<16>, b0, b1, rows, grid, it), f = run(p2<18>, b0, b1, rows, grid, it);
        const float g = run(p0<0>, b0, b1, rows, grid, it), h = run(p1<0>, b0, b1, rows, grid, it);
        const int g3 = (int)((rows * D * 4 + 16383) / 16384);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&p4<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        printf("%7zu envs 16 KiB workgroups: sc1 %.2f plain %.2f nt %.2f | own-region sc1 at 8 waves/CU %.2f, 4 waves/CU %.2f\n", envs,
               mb / run(p3<16>, b0, b1, rows, g3, it), mb / run(p3<0>, b0, b1, rows, g3, it), mb / run(p3<2>, b0, b1, rows, g3, it),
               mb / run(p4<16>, b0, b1, rows, grid, it, 60 * 1024), mb / run(p4<16>, b0, b1, rows, grid, it, 100 * 1024));
        printf("%7zu envs %6.0f MB  TB/s: own-region sc1 %.2f nt+sc1 %.2f plain %.2f | workgroup-dealt sc1 %.2f nt+sc1 %.2f plain %.2f | 64B-per-lane sc1 %.2f nt+sc1 %.2f\n",
               envs, mb, mb / a, mb / b, mb / g, mb / c, mb / d, mb / h, mb / e, mb / f);
        hipFree(b0); hipFree(b1);
    }
    return 0;
}

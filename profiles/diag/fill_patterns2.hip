// Second look at store patterns beyond the last-level cache: is it the instruction (buffer store
// with a descriptor against a plain global store), the bytes a workgroup writes before it ends,
// or the order in which the chip as a whole walks through memory?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_patterns2 profiles/diag/fill_patterns2.hip && /tmp/fill_patterns2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// a workgroup writes BYTES consecutive bytes with plain global stores and ends
template <int BYTES>
__global__ void __launch_bounds__(256) short_wg(float* out, size_t nbytes) {
    const size_t b0 = (size_t)blockIdx.x * BYTES;
    f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
#pragma unroll
    for (int j = 0; j < BYTES / 4096; ++j) {
        const size_t off = b0 + (size_t)(j * 256 + threadIdx.x) * 16;
        if (off < nbytes) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + off) = v;
    }
}

// every wave streams its own 51.7 KB region front to back, plain global stores
__global__ void __launch_bounds__(256) own_region_global(float* out, size_t nbytes) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t base = wave * 51712;
    f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
#pragma unroll
    for (int k = 0; k < 51; ++k) {
        const size_t off = base + (size_t)k * 1024 + lane * 16;
        if (off < base + 51712 && off < nbytes) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + off) = v;
    }
}

// 4 KiB per workgroup like short_wg<4096>, but workgroup b writes chunk b + SHIFT (within its group of
// 64 chunks): with workgroups dealt round robin over the 8 XCDs, SHIFT moves every XCD's chunks to
// another residue class modulo 8
template <int SHIFT>
__global__ void __launch_bounds__(256) short_wg_shifted(float* out, size_t nbytes) {
    const size_t b = blockIdx.x;
    const size_t chunk = (b & ~(size_t)63) | ((b + SHIFT) & 63);
    const size_t off = chunk * 4096 + (size_t)threadIdx.x * 16;
    f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    if (off < nbytes) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + off) = v;
}

// own region, but a wave never has more than LIMIT + 1 stores in flight
template <int LIMIT>
__global__ void __launch_bounds__(256) own_region_paced(float* out, size_t nbytes) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t base = wave * 51712;
    f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
#pragma unroll
    for (int k = 0; k < 51; ++k) {
        const size_t off = base + (size_t)k * 1024 + lane * 16;
        if (off < base + 51712 && off < nbytes) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + off) = v;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LIMIT) : "memory");
    }
}

// own region, but wave w starts writing only when at most WINDOW waves below it are unfinished:
// the chip's write front stays WINDOW x 51.7 KB wide.  Bounded spin (a wave that waited too long
// goes ahead), the counter wraps to zero when the last wave has signed off.
__device__ unsigned int g_done;
template <int WINDOW>
__global__ void __launch_bounds__(256) own_region_windowed(float* out, size_t nbytes) {
    const int lane = threadIdx.x & 63;
    const unsigned int wave = blockIdx.x * 4 + (threadIdx.x >> 6), total = gridDim.x * 4;
    int spins = 0;
    while ((int)(wave - __hip_atomic_load(&g_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) > WINDOW && ++spins < 20000)
        __builtin_amdgcn_s_sleep(8);
    const size_t base = (size_t)wave * 51712;
    f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
#pragma unroll
    for (int k = 0; k < 51; ++k) {
        const size_t off = base + (size_t)k * 1024 + lane * 16;
        if (off < base + 51712 && off < nbytes) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + off) = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) atomicInc(&g_done, total - 1);
}

// long-lived workgroups, the whole grid sweeps memory together: iteration i of workgroup b writes
// the 4 KiB chunk i * gridDim + b
__global__ void __launch_bounds__(256) grid_stride(float* out, size_t nbytes) {
    f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    for (size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; off < nbytes; off += (size_t)gridDim.x * 4096)
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + off) = v;
}

template <class K>
static double run(K kernel, float* b0, float* b1, size_t nbytes, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, i & 1 ? b1 : b0, nbytes);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, i & 1 ? b1 : b0, nbytes);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)nbytes / 1e6 / (ms * 1000.0 / iters);  // TB/s
}

int main() {
    for (size_t envs : {65536ul, 131072ul, 524288ul}) {
        const size_t nbytes = envs * 2 * 202 * 4;
        float *b0, *b1;
        if (hipMalloc(&b0, nbytes) != hipSuccess || hipMalloc(&b1, nbytes) != hipSuccess) return 1;
        const int it = 60;
        printf("%4zu MB  short workgroups 4K %.2f  8K %.2f  16K %.2f  64K %.2f | own region (global stores) %.2f | grid-stride 2048 wg %.2f  8192 wg %.2f  512 wg %.2f TB/s\n",
               nbytes / 1000000,
               run(short_wg<4096>, b0, b1, nbytes, (int)((nbytes + 4095) / 4096), it),
               run(short_wg<8192>, b0, b1, nbytes, (int)((nbytes + 8191) / 8192), it),
               run(short_wg<16384>, b0, b1, nbytes, (int)((nbytes + 16383) / 16384), it),
               run(short_wg<65536>, b0, b1, nbytes, (int)((nbytes + 65535) / 65536), it),
               run(own_region_global, b0, b1, nbytes, (int)((nbytes / 51712 + 3) / 4), it),
               run(grid_stride, b0, b1, nbytes, 2048, it), run(grid_stride, b0, b1, nbytes, 8192, it),
               run(grid_stride, b0, b1, nbytes, 512, it));
        const int g = (int)((nbytes / 51712 + 3) / 4);
        printf("%4zu MB  own region, stores in flight per wave <= 1: %.2f  2: %.2f  3: %.2f  5: %.2f  9: %.2f  17: %.2f TB/s\n", nbytes / 1000000,
               run(own_region_paced<0>, b0, b1, nbytes, g, it), run(own_region_paced<1>, b0, b1, nbytes, g, it),
               run(own_region_paced<2>, b0, b1, nbytes, g, it), run(own_region_paced<4>, b0, b1, nbytes, g, it),
               run(own_region_paced<8>, b0, b1, nbytes, g, it), run(own_region_paced<16>, b0, b1, nbytes, g, it));
        const int g4 = (int)((nbytes + 4095) / 4096);
        printf("%4zu MB  4 KiB workgroups, chunk index shifted by 0: %.2f  1: %.2f  2: %.2f  4: %.2f  8: %.2f  16: %.2f TB/s\n", nbytes / 1000000,
               run(short_wg_shifted<0>, b0, b1, nbytes, g4, it), run(short_wg_shifted<1>, b0, b1, nbytes, g4, it),
               run(short_wg_shifted<2>, b0, b1, nbytes, g4, it), run(short_wg_shifted<4>, b0, b1, nbytes, g4, it),
               run(short_wg_shifted<8>, b0, b1, nbytes, g4, it), run(short_wg_shifted<16>, b0, b1, nbytes, g4, it));
        printf("%4zu MB  own region, write front limited to 64 waves: %.2f  160: %.2f  320: %.2f  640: %.2f  1280: %.2f  2560: %.2f TB/s\n", nbytes / 1000000,
               run(own_region_windowed<64>, b0, b1, nbytes, g, it), run(own_region_windowed<160>, b0, b1, nbytes, g, it),
               run(own_region_windowed<320>, b0, b1, nbytes, g, it), run(own_region_windowed<640>, b0, b1, nbytes, g, it),
               run(own_region_windowed<1280>, b0, b1, nbytes, g, it), run(own_region_windowed<2560>, b0, b1, nbytes, g, it));
        hipFree(b0); hipFree(b1);
    }
    return 0;
}

// One store per wave (256-lane workgroups, 4 KiB each, sc1), but workgroup b writes chunk perm(b):
// consecutive workgroups, which run at the same time, are SPREAD apart, so the chip writes all over
// the tensor at any moment (a wide front) while every wave still issues a single store.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) fill_perm(float* out, uint32_t nchunks, uint32_t ways) {
    // chunk = (b % ways) * (nchunks / ways) + b / ways: `ways` interleaved sequential streams
    const uint32_t b = blockIdx.x;
    const uint32_t chunk = (b % ways) * (nchunks / ways) + b / ways;
    char* base = reinterpret_cast<char*>(out) + (size_t)chunk * 4096;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 4096, 0x00020000);
    i32x4 v = {1, 2, 3, 4};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, threadIdx.x * 16, 0, 16);
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
    const uint32_t nchunks = 1u << 18;  // 1 GiB
    const size_t nbytes = (size_t)nchunks * 4096;
    float *b0, *b1;
    if (hipMalloc(&b0, nbytes) != hipSuccess || hipMalloc(&b1, nbytes) != hipSuccess) return 1;
    for (uint32_t ways : {1u, 2u, 8u, 64u, 512u, 2048u, 16384u}) {
        const double t = run([&](int i) { hipLaunchKernelGGL(fill_perm, dim3(nchunks), dim3(256), 0, 0, i & 1 ? b1 : b0, nchunks, ways); }, 30);
        printf("%5u interleaved sequential streams of 4 KiB chunks, one store per wave: %.2f TB/s\n", ways, nbytes / 1e6 / t);
    }
    return 0;
}

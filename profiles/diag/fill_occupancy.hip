// The fast pattern (256-lane workgroups, one 16-byte store per lane, address order, sc1) with its
// occupancy cut by a dynamic LDS allocation, and the slow one (own 51,712-byte region per wave) with
// many more waves per CU (one-wave workgroups with regions of 1/4 and 1/13 the size): is the number of
// resident waves what sets the number of writes in flight?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
extern __shared__ char dyn[];

__global__ void __launch_bounds__(256) fill_short(float* out) {
    char* base = reinterpret_cast<char*>(out) + (size_t)blockIdx.x * 4096;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 4096, 0x00020000);
    i32x4 v = {1, 2, 3, 4};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, threadIdx.x * 16, 0, 16);
}

// a workgroup of 256 lanes writes NCHUNK consecutive 4 KiB chunks, one store per lane per chunk, and ends
template <int NCHUNK>
__global__ void __launch_bounds__(256) fill_multi(float* out) {
    char* base = reinterpret_cast<char*>(out) + (size_t)blockIdx.x * 4096 * NCHUNK;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 4096 * NCHUNK, 0x00020000);
    i32x4 v = {1, 2, 3, 4};
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, c * 4096 + threadIdx.x * 16, 0, 16);
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
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_short), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned gs = (unsigned)(nbytes / 4096);
    for (int lds_kb : {0, 16, 32, 64, 128}) {
        const double t = run([&](int i) { hipLaunchKernelGGL(fill_short, dim3(gs), dim3(256), (size_t)lds_kb * 1024, 0, i & 1 ? b1 : b0); }, 40);
        printf("one store per wave, %3d KiB LDS per workgroup (%s workgroups per CU): %.2f TB/s\n", lds_kb,
               lds_kb == 0 ? "8" : lds_kb == 16 ? "8" : lds_kb == 32 ? "5" : lds_kb == 64 ? "2" : "1", nbytes / 1e6 / t);
    }
    printf("256-lane workgroups writing 2 / 4 / 13 consecutive 4 KiB chunks (2 / 4 / 13 stores per wave): %.2f / %.2f / %.2f TB/s\n",
           nbytes / 1e6 / run([&](int i) { hipLaunchKernelGGL(fill_multi<2>, dim3(gs / 2), dim3(256), 0, 0, i & 1 ? b1 : b0); }, 40),
           nbytes / 1e6 / run([&](int i) { hipLaunchKernelGGL(fill_multi<4>, dim3(gs / 4), dim3(256), 0, 0, i & 1 ? b1 : b0); }, 40),
           nbytes / 1e6 / run([&](int i) { hipLaunchKernelGGL(fill_multi<13>, dim3(gs / 13), dim3(256), 0, 0, i & 1 ? b1 : b0); }, 40));
    return 0;
}

// The env kernel's store stream (a 51.7 KB region per wave, 51 stores of 1 KiB, 2048 waves) into
// memory of different kinds: plain hipMalloc, hipExtMallocWithFlags uncached / fine-grained.
// Ten 106 MB targets in turn (HBM regime).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 51;

template <int AUX>
__global__ void __launch_bounds__(256) fill(char* out, uint32_t W) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    char* base = out + (size_t)w * K * 1024;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(((uint64_t)hi << 32) | lo), 0, K * 1024, 0x00020000);
    i32x4 v = {(int)w, 2, 3, 4};
#pragma unroll
    for (int k = 0; k < K; ++k) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane * 16 + k * 1024, 0, AUX);
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
    struct Kind { const char* name; int mode; } kinds[] = {{"hipMalloc", 0}, {"uncached", 1}, {"fine-grained", 2}};
    for (auto kind : kinds) {
        char* buf[NB];
        bool ok = true;
        for (auto& b : buf) {
            hipError_t e = kind.mode == 0 ? hipMalloc((void**)&b, nbytes)
                         : hipExtMallocWithFlags((void**)&b, nbytes, kind.mode == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
            if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", kind.name, hipGetErrorString(e)); ok = false; break; }
        }
        if (!ok) continue;
        const double t0 = run([&](int i) { hipLaunchKernelGGL(fill<16>, dim3(W / 4), dim3(256), 0, 0, buf[i % NB], W); }, 60);
        const double t1 = run([&](int i) { hipLaunchKernelGGL(fill<18>, dim3(W / 4), dim3(256), 0, 0, buf[i % NB], W); }, 60);
        const double t2 = run([&](int i) { hipLaunchKernelGGL(fill<0>, dim3(W / 4), dim3(256), 0, 0, buf[i % NB], W); }, 60);
        printf("%-13s region per wave: sc1 %.2f us (%.2f TB/s), sc1+nt %.2f us (%.2f TB/s), plain %.2f us (%.2f TB/s)\n", kind.name, t0,
               nbytes / 1e6 / t0, t1, nbytes / 1e6 / t1, t2, nbytes / 1e6 / t2);
        for (auto& b : buf) (void)hipFree(b);
    }
    return 0;
}

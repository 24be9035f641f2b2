// Many stores per wave, but a compact front: 2048 resident waves (8 per CU), each issues 51 stores
// of 1 KiB; store k of wave w goes to chunk k * W + w, so that waves running in step sweep the
// tensor like a fill does.  Against the env kernel's ownership (store k of wave w at chunk
// 51 w + k).  `spin` delays each wave by a different short time first (the simulation's jitter).
// Not realisable with rows owned by waves -- it separates "many stores per wave" from "wide front".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 51;

template <bool INTERLEAVED>
__global__ void __launch_bounds__(256) fill(char* out, uint32_t W, uint32_t spin) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t x = w * 2654435761u;
    for (uint32_t k = 0; k < (x >> 28) * spin; ++k) x = x * 1664525u + 1013904223u;
    i32x4 v = {(int)x, 2, 3, 4};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t chunk = INTERLEAVED ? (size_t)k * W + w : (size_t)w * K + k;
        __builtin_nontemporal_store(v, reinterpret_cast<i32x4*>(out + chunk * 1024) + lane);
    }
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
    for (auto& b : buf) if (hipMalloc(&b, nbytes) != hipSuccess) return 1;
    for (uint32_t spin : {0u, 8u, 32u}) {
        const double t0 = run([&](int i) { hipLaunchKernelGGL(fill<false>, dim3(W / 4), dim3(256), 0, 0, buf[i % NB], W, spin); }, 60);
        const double t1 = run([&](int i) { hipLaunchKernelGGL(fill<true>, dim3(W / 4), dim3(256), 0, 0, buf[i % NB], W, spin); }, 60);
        printf("spin %2u: region per wave %.2f us (%.2f TB/s), interleaved %.2f us (%.2f TB/s)\n", spin, t0, nbytes / 1e6 / t0, t1,
               nbytes / 1e6 / t1);
    }
    return 0;
}

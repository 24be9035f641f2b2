// The env kernel's store stream (a 51.7 KB region per wave, 2048 waves, ten 106 MB targets in turn)
// with 4, 8 or 16 bytes per lane and store: 256 B, 512 B or 1 KiB per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
constexpr int REGION = 51 * 1024;

template <int BYTES, int AUX>
__global__ void __launch_bounds__(256) fill(char* out, uint32_t W) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    char* base = out + (size_t)w * REGION;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(((uint64_t)hi << 32) | lo), 0, REGION, 0x00020000);
    constexpr int PER = 64 * BYTES, N = REGION / PER;
#pragma unroll 8
    for (int k = 0; k < N; ++k) {
        const uint32_t off = lane * BYTES + k * PER;
        if constexpr (BYTES == 16) __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)w, 2, 3, 4}, rsrc, off, 0, AUX);
        else if constexpr (BYTES == 8) __builtin_amdgcn_raw_buffer_store_b64(i32x2{(int)w, 2}, rsrc, off, 0, AUX);
        else __builtin_amdgcn_raw_buffer_store_b32((int)w, rsrc, off, 0, AUX);
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
    const size_t nbytes = (size_t)W * REGION;
    constexpr int NB = 10;
    char* buf[NB];
    for (auto& b : buf) if (hipMalloc((void**)&b, nbytes) != hipSuccess) return 1;
#define CASE(BYTES, AUX) { const double t = run([&](int i) { hipLaunchKernelGGL((fill<BYTES, AUX>), dim3(W / 4), dim3(256), 0, 0, buf[i % NB], W); }, 60); \
        printf("%2d bytes per lane, aux %2d: %.2f us (%.2f TB/s)\n", BYTES, AUX, t, nbytes / 1e6 / t); }
    CASE(16, 16) CASE(8, 16) CASE(4, 16)
    CASE(16, 18) CASE(8, 18) CASE(4, 18)
    CASE(16, 0) CASE(8, 0) CASE(4, 0)
    return 0;
}

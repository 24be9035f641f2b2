// What does the widening kernel (int8 -> float32, short workgroups) cost as a function of where its
// input comes from?  Output: a ring of 8 x 106 MB (rows kept, HBM regime).  Input: (a) a different
// 26.5 MB buffer every launch out of 8 (cold), (b) the same 26.5 MB buffer every launch (last-level
// cache), (c) 1 MB read over and over (L2), (d) no input at all (a fill with the same geometry).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/expand_src profiles/diag/expand_src.hip && /tmp/expand_src
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, int MODE>  // MODE 0: src[g], 1: src[g & mask], 2: no load
__global__ void __launch_bounds__(256) expand(const uint32_t* __restrict__ img, float* __restrict__ out, uint32_t nwords, uint32_t mask) {
    const uint32_t g0 = blockIdx.x * (uint32_t)(256 * U) + threadIdx.x;
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t g = g0 + (uint32_t)(u * 256);
        v[u] = MODE == 2 ? g : (g < nwords ? __builtin_nontemporal_load(img + (MODE == 1 ? (g & mask) : g)) : 0u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t g = g0 + (uint32_t)(u * 256);
        if (g < nwords) {
            f32x4 f;
            f.x = (float)(int)(int8_t)(v[u]); f.y = (float)(int)(int8_t)(v[u] >> 8);
            f.z = (float)(int)(int8_t)(v[u] >> 16); f.w = (float)(int)(int8_t)(v[u] >> 24);
            __builtin_nontemporal_store(f, reinterpret_cast<f32x4*>(out) + g);
        }
    }
}

int main() {
    const size_t nwords = (size_t)65536 * 2 * 202 / 4, out_bytes = nwords * 16, in_bytes = nwords * 4;
    float* out[8]; uint32_t* in[8];
    for (int i = 0; i < 8; ++i) {
        if (hipMalloc(&out[i], out_bytes) != hipSuccess || hipMalloc(&in[i], in_bytes) != hipSuccess) return 1;
        hipMemset(in[i], 1, in_bytes);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kernel, int U, int src_mode, uint32_t mask, const char* name) {
        const int grid = (int)((nwords + 256 * U - 1) / (256 * U));
        for (int rep = 0; rep < 2; ++rep) {
            if (rep) hipEventRecord(e0, 0);
            for (int i = 0; i < 64; ++i)
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, in[src_mode == 0 ? i % 8 : 0], out[i % 8], (uint32_t)nwords, mask);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %6.2f us per launch  %.2f TB/s written\n", name, ms * 1000 / 64, out_bytes / 1e6 / (ms * 1000 / 64));
    };
    run(expand<1, 2>, 1, 1, 0, "no input, 1 store per lane");
    run(expand<2, 2>, 2, 1, 0, "no input, 2 stores per lane");
    run(expand<1, 1>, 1, 1, 0x3FFFF, "1 MB input (L2), 1 dword per lane");
    run(expand<2, 1>, 2, 1, 0x3FFFF, "1 MB input (L2), 2 dwords per lane");
    run(expand<1, 0>, 1, 1, 0, "same 26.5 MB input every launch, 1 dword per lane");
    run(expand<2, 0>, 2, 1, 0, "same 26.5 MB input every launch, 2 dwords per lane");
    run(expand<1, 0>, 1, 0, 0, "8 inputs rotated (cold), 1 dword per lane");
    run(expand<2, 0>, 2, 0, 0, "8 inputs rotated (cold), 2 dwords per lane");
    return 0;
}

// The env kernel's store stream with two assignments of envs to waves.  Every wave writes 32 envs
// of 1616 bytes (2 rows of 202 floats) with 51 stores of 64 x 16 bytes, as the kernel does:
//   region : wave w owns envs 32 w .. 32 w + 31 (one contiguous 51.7 KB region per wave);
//   strided: a wave's 32 envs are 32 / GE groups of GE consecutive envs, group j of wave w being
//            group number w + j W of the tensor (W = number of waves): the j-th groups of all the
//            waves together are one contiguous stretch, so waves that run in step write few
//            sequential streams at any moment.  GE = 8 makes a group a whole number of 128-byte
//            lines (8 x 1616 = 101 x 128); with GE = 1 neighbouring waves share the lines at both
//            ends of every env.
// Ten tensors of 106 MB are written in turn, so that every launch goes to HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t ENV_BYTES = 1616, CPE = 101, EPW = 32, ITERS = (EPW * CPE + 63) / 64;

template <int K>
struct Step {
    template <class F>
    static __device__ __forceinline__ void run(F&& f) {
        f(std::integral_constant<int, K>{});
        if constexpr (K + 1 < (int)ITERS) Step<K + 1>::run(f);
    }
};

template <int GE, int AUX>
__global__ void __launch_bounds__(256) fill_envs(char* out, uint32_t E, uint32_t W, uint32_t spin) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= W) return;
    // stand-in for the simulation: waves reach the stream at slightly different times
    uint32_t x = w * 2654435761u;
    for (uint32_t k = 0; k < spin + (x >> 28) * (spin >> 3); ++k) x = x * 1664525u + 1013904223u;
    constexpr uint32_t CPG = CPE * GE;  // 16-byte chunks per group
    const size_t first_env = (size_t)w * GE;
    const uint32_t gap = (W - 1) * GE * ENV_BYTES;  // extra distance between this wave's consecutive groups
    const uint32_t nj = EPW / GE;                   // (E is a multiple of 32 W here)
    char* base = out + first_env * ENV_BYTES;
    const uint32_t wb_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
    const uint32_t wb_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
    char* ubase = reinterpret_cast<char*>(((uint64_t)wb_hi << 32) | wb_lo);
    const uint32_t range = (uint32_t)((size_t)E * ENV_BYTES - first_env * ENV_BYTES);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, __builtin_amdgcn_readfirstlane(range), 0x00020000);
    i32x4 v = {(int)x, 2, 3, 4};
    Step<0>::run([&](auto K) {
        constexpr uint32_t k = K.value;
        constexpr uint32_t q = (k * 64) / CPG;
        constexpr uint32_t b = (q + 1) * CPG - k * 64;  // first lane that belongs to group q + 1
        uint32_t el = q;
        if constexpr (b < 64) el = lane >= b ? q + 1 : q;
        uint32_t voff = lane * 16u + k * 1024u + el * gap;
        voff = el < nj ? voff : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, AUX);
    });
}

template <class F>
static double run(F launch, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 12; ++i) launch(i);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch(i);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / iters;
}

int main() {
    const uint32_t E = 65536, W = E / EPW;
    const size_t nbytes = (size_t)E * ENV_BYTES;
    constexpr int NB = 10;
    char* buf[NB];
    for (auto& b : buf) if (hipMalloc(&b, nbytes) != hipSuccess) return 1;
    for (uint32_t spin : {0u, 64u}) {
        const dim3 grid(W / 4), block(256);
#define CASE(GE, AUX) { const double t = run([&](int i) { hipLaunchKernelGGL((fill_envs<GE, AUX>), grid, block, 0, 0, buf[i % NB], E, W, spin); }, 60); \
            printf("spin %3u, groups of %2d envs, aux %2d: %.2f us (%.2f TB/s)\n", spin, GE, AUX, t, nbytes / 1e6 / t); }
        CASE(32, 18) CASE(16, 18) CASE(8, 18) CASE(1, 18)
        CASE(32, 16) CASE(16, 16) CASE(8, 16) CASE(1, 16)
    }
    // verify the strided mapping covers every byte exactly: fill with zeros, run, count
    hipMemset(buf[0], 0, nbytes);
    hipLaunchKernelGGL((fill_envs<8, 16>), dim3(W / 4), dim3(256), 0, 0, buf[0], E, W, 0u);
    hipDeviceSynchronize();
    int* host = (int*)malloc(nbytes);
    hipMemcpy(host, buf[0], nbytes, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < nbytes / 16; ++i) if (host[i * 4 + 1] != 2) ++bad;
    printf("strided mapping: %zu of %zu chunks not written\n", bad, nbytes / 16);
    return 0;
}

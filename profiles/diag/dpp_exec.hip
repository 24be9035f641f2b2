// What does a DPP quad_perm read return for a source lane that EXEC has switched off (bound_ctrl:1)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_exec profiles/diag/dpp_exec.hip && /tmp/dpp_exec
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    int x = 1000 + lane;          // every lane's register holds 1000 + lane
    int got = -1;
    if ((lane & 3) != 3) {        // lane 3 of every quad is switched off from here on
        x = lane;                 // the active lanes overwrite theirs; lane 3's copy stays 1000 + lane
        got = __builtin_amdgcn_mov_dpp(x, 0xFF /* quad_perm [3,3,3,3] */, 0xF, 0xF, true);
    }
    out[lane] = got;
}

// the quad-wide OR as the env kernel forms it, with lane 3 of every quad switched off and a
// stale non-zero value left in its register
__global__ void probe_or(int* out, const int* in) {
    const int lane = threadIdx.x;
    unsigned x = 77u + in[lane];  // junk in every lane, lane 3 keeps it
    int got = -1;
    if ((lane & 3) != 3) {
        x = ((lane & 3) == 1) ? 1u : 0u;
        x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
        x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);
        got = (int)x;
    }
    out[lane] = got;
}

// the same OR written as the two VOP2 DPP instructions the compiler emits for it, on a register
// that holds 1000 + lane in the switched-off lane
__global__ void probe_or_vop2(int* out) {
    const int lane = threadIdx.x;
    unsigned x = 1000u + lane;
    asm volatile("" : "+v"(x));
    int got = -1;
    if ((lane & 3) != 3) {
        const unsigned y = ((lane & 3) == 1) ? 1u : 0u;
        asm volatile("v_mov_b32 %0, %1\n\ts_nop 1\n\t"
                     "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                     "v_or_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(x) : "v"(y));
        got = (int)x;
    }
    out[lane] = got;
}

int main() {
    int* d;
    int h[64];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("lanes 0..7 read from lane 3 / 7 of their quad: ");
    for (int i = 0; i < 8; ++i) printf("%d ", h[i]);
    printf("\n(0 = switched-off lanes read as zero; 1003 / 1007 = their stale register contents)\n");
    int* din;
    hipMalloc(&din, sizeof h);
    hipMemset(din, 0, sizeof h);
    hipLaunchKernelGGL(probe_or, dim3(1), dim3(64), 0, 0, d, din);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("quad-wide OR of (0, 1, 0, off): ");
    for (int i = 0; i < 8; ++i) printf("%d ", h[i]);
    printf("\n(1 1 1 -1 expected if switched-off lanes read as zero)\n");
    hipLaunchKernelGGL(probe_or_vop2, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("the same with v_or_b32_dpp on a register whose switched-off lane holds 1000 + lane: ");
    for (int i = 0; i < 8; ++i) printf("%d ", h[i]);
    printf("\n(1 1 1 -1 if the switched-off lane reads as zero; 1003-ish values if its register is read)\n");
    return 0;
}

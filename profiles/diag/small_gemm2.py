"""Second policy layer (rows x 512 -> 256, fp32) at small row counts: the fused hipBLASLt call
against a plain rocBLAS product followed by in-place bias add and ReLU.  us per call (graph replay)."""
import torch

dev = "cuda"


def graph_timed(fn, n=400):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n // 20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n // 20 * 20)


import sys
CASES = [((512, 256), (4, 32, 64, 128, 256, 512, 1024)), ((202, 512), (4, 32, 64, 128, 256, 512, 1024))]
if len(sys.argv) > 1:  # the centralised critic's layers (one row) and odd widths
    CASES = [((6464, 512), (1, 2)), ((512, 256), (1, 2, 8, 16, 300, 384, 448)), ((256, 1), (1, 32)), ((420, 512), (64, 128)), ((105, 512), (64, 128)),
             ((256, 8), (32, 128, 256))]
for (K, M), Ns in CASES:
    for N in Ns:
        x = torch.randn(N, K, device=dev); W = torch.randn(M, K, device=dev); b = torch.randn(M, device=dev)
        torch.backends.cuda.preferred_blas_library("hipblaslt")
        fused = graph_timed(lambda: torch._addmm_activation(b, x, W.t(), use_gelu=False))
        torch.backends.cuda.preferred_blas_library("hipblas")
        plain = graph_timed(lambda: torch.mm(x, W.t()).add_(b).relu_())
        only = graph_timed(lambda: torch.mm(x, W.t()))
        torch.backends.cuda.preferred_blas_library("hipblaslt")
        print(f"K={K} M={M} N={N:5d}: fused hipBLASLt {fused:6.2f} us | rocBLAS mm + add_ + relu_ {plain:6.2f} us (mm alone {only:5.2f})")

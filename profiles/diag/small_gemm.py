"""First policy layer at the reference's batch sizes (N rows x 202 -> 512, fp32): how long do the
ways PyTorch offers to compute relu(x W^T + b) take?  us per call, 300 calls between HIP events."""
import sys

import torch
import torch.nn.functional as F

dev = "cuda"
torch.manual_seed(0)


def timed(fn, n=300):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def graph_timed(fn, n=300):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    return timed(g.replay, n // 20 + 1) / 20


for N in [int(a) for a in sys.argv[1:]] or [32, 512, 8192]:
    for (K, M) in ((202, 512), (512, 256), (256, 8)):
        x = torch.randn(N, K, device=dev)
        W = torch.randn(M, K, device=dev) / K ** 0.5
        b = torch.randn(M, device=dev)
        Wt = W.t().contiguous()
        xp = F.pad(x, (0, 256 - K)) if K < 256 else x
        Wp = F.pad(W, (0, 256 - K)) if K < 256 else W
        out = {}
        for lib in ("hipblaslt", "hipblas"):
            torch.backends.cuda.preferred_blas_library(lib)
            out[f"{lib}: _addmm_activation(b, x, W.t())"] = graph_timed(lambda: torch._addmm_activation(b, x, W.t(), use_gelu=False))
            out[f"{lib}: addmm(b, x, W.t())"] = graph_timed(lambda: torch.addmm(b, x, W.t()))
            out[f"{lib}: addmm(b, x, Wt) (W stored transposed)"] = graph_timed(lambda: torch.addmm(b, x, Wt))
            out[f"{lib}: mm(x, W.t())"] = graph_timed(lambda: torch.mm(x, W.t()))
            out[f"{lib}: mm(W, x.t()) (transposed product)"] = graph_timed(lambda: torch.mm(W, x.t()))
            if K < 256:
                out[f"{lib}: addmm on K padded to 256"] = graph_timed(lambda: torch.addmm(b, xp, Wp.t()))
        torch.backends.cuda.preferred_blas_library("hipblaslt")
        print(f"N={N} K={K} M={M}")
        for k, v in out.items():
            print(f"   {v:7.2f} us  {k}")

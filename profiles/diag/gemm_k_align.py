"""Does the first policy layer's K = 202 (rows of 808 bytes, not 16-byte aligned) cost the GEMM anything?
Times relu(x @ W^T + b) for x [rows, K] -> 512 with K = 202 / 204 / 208 (contiguous) and for a
[rows, 202] view of a [rows, 208] tensor, plus the weight-gradient product of the backward pass."""
import torch

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

dev = "cuda"
for rows in (131072, 262144, 16384):
    for K in (202, 204, 208, 256):
        x = torch.randint(-3, 4, (rows, K), device=dev).float()
        w = torch.randn(512, K, device=dev); b = torch.randn(512, device=dev)
        gy = torch.randn(rows, 512, device=dev)
        f = timed(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))
        g = timed(lambda: gy.t() @ x)
        print(f"rows {rows} K {K}: forward {f:7.1f} us ({2*rows*K*512/f/1e6:6.1f} TFLOP/s)  dW {g:7.1f} us ({2*rows*K*512/g/1e6:6.1f} TFLOP/s)", flush=True)
    xp = torch.randint(-3, 4, (rows, 208), device=dev).float()
    xv = xp[:, :202]
    w = torch.randn(512, 202, device=dev); b = torch.randn(512, device=dev)
    f = timed(lambda: torch.addmm(b, xv, w.t()))
    print(f"rows {rows} K 202 as a view of 208-wide rows: forward (addmm) {f:7.1f} us", flush=True)

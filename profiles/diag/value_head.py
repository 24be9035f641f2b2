"""Cost of the critic head (hidden[N,256] -> value[N]) in a launch-bound rollout step, replayed
from a hipGraph of 64 calls: what torch offers for a one-column product."""
import torch
dev = "cuda"
for N in (512, 8192):
    h = torch.randn(N, 256, device=dev); W = torch.randn(1, 256, device=dev); b = torch.randn(1, device=dev)
    row = torch.empty(N, device=dev)
    variants = {
        "addmm(out=view)": lambda: torch.addmm(b, h, W.t(), out=row.view(-1, 1)),
        "addmv(out)": lambda: torch.addmv(b, h, W.view(-1), out=row),
        "mv(out)+add_": lambda: torch.mv(h, W.view(-1), out=row).add_(b),
        "mul+sum": lambda: torch.sum(h * W, dim=1, out=row).add_(b),
    }
    ref = (h @ W.t() + b).squeeze(1)
    for name, fn in variants.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        assert torch.allclose(row, ref, rtol=1e-4, atol=1e-4), name
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(64):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"N={N} {name:18s} {e0.elapsed_time(e1) * 1e3 / (20 * 64):.2f} us per call")

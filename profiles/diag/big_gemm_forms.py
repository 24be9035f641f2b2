"""The policy's two trunk layers at rollout size (131,072 rows, fp32): hipBLASLt's fused
bias+ReLU GEMM (what the product runs) against a rocBLAS product plus in-place bias and ReLU."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd import policy as P

dev = torch.device("cuda")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
for (k, n) in ((202, 512), (512, 256), (256, 8)):
    x = torch.randn(rows, k, device=dev)
    w = torch.randn(n, k, device=dev)
    b = torch.randn(n, device=dev)
    for name in ("fused", "plain"):
        fn = P._FORMS[name]
        for _ in range(3):
            fn(x, w, b, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(x, w, b, True)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{rows} x {k} -> {n}: {name:5s} {us:8.1f} us  {2 * rows * k * n / us / 1e6:6.1f} TFLOP/s", flush=True)

"""What does a plain device fill reach on this GPU at sizes beyond the last-level cache?"""
import torch
for mb in (106, 212, 424, 848, 1696):
    n = mb * 1000 * 1000 // 4
    x = [torch.empty(n, device="cuda") for _ in range(2)]
    for i in range(4): x[i % 2].fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(50): x[i % 2].fill_(float(i))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    z = [torch.zeros(n, device="cuda") for _ in range(2)]
    e0.record()
    for i in range(50): z[i % 2].zero_()
    e1.record(); torch.cuda.synchronize()
    us2 = e0.elapsed_time(e1) * 20
    print("%5d MB: fill_ %.1f us = %.2f TB/s | zero_ (memset) %.1f us = %.2f TB/s" % (mb, us, n * 4 / us / 1e6, us2, n * 4 / us2 / 1e6))

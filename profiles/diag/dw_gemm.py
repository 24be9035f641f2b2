import torch
torch.manual_seed(0)
def timeit(f, n=20):
    for _ in range(3): f()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for K in (262144, 65536, 16384):
  for (i,o) in ((202,512),(512,256),(256,7)):
    x=torch.randn(K,i,device="cuda"); dy=torch.randn(K,o,device="cuda")
    ref = dy.t() @ x
    t0 = timeit(lambda: dy.t() @ x)
    t1 = timeit(lambda: (x.t() @ dy).t())
    line = "K=%d %dx%d  dy.t()@x %.0f us | (x.t()@dy).t() %.0f us" % (K,o,i,t0,t1)
    for S in (8,32,128):
        f = lambda: torch.bmm(dy.view(S, K//S, o).transpose(1,2), x.view(S, K//S, i)).sum(0)
        err = (f()-ref).abs().max().item() / ref.abs().max().item()
        line += " | bmm S=%d %.0f us (rel %.1e)" % (S, timeit(f), err)
    print(line)

import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import torch
    sys.path.insert(0, os.getcwd())
    from marl_amd.gae import compute_gae
    T, N = 128, 131072
    r = torch.rand(T, N, device="cuda"); d = (torch.rand(T, N, device="cuda") < 0.01).float()
    v = torch.randn(T, N, device="cuda"); nv = torch.randn(1, N, device="cuda")
    for _ in range(10): compute_gae(r, d, v, nv, 0.99, 0.95)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): compute_gae(r, d, v, nv, 0.99, 0.95)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print("%s: %.1f us  %.0f GB/s" % (os.environ.get("MARL_AMD_LIB", "default").split("libvar_")[-1], us, T * N * 20 / us / 1e3))
else:
    for rep in range(2):
        for name in sys.argv[1:]:
            env = dict(os.environ, MARL_AMD_LIB=os.path.join(os.getcwd(), "profiles/diag/libvar_%s.so" % name))
            subprocess.run([sys.executable, __file__, "one"], env=env, check=True)

"""Per-phase cycle counts and wall-clock timeline of the env kernel for any config, from the
-DOC_STAMPS diagnostic library:
    sh profiles/diag/build_variants.sh stamps:-DOC_STAMPS
    MARL_AMD_LIB=$PWD/profiles/diag/libvar_stamps.so python profiles/diag/stamps_case.py <layout> <agents> <feature> <envs>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv  # noqa: E402
from marl_amd.layouts import get_layout
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _stamped import stamped_step  # noqa: E402

layout, A, feature, E = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
env = BatchedOvercookedVecEnv(get_layout(layout), A, feature, E, seed=1)
N, D = env.N, env.D
obs = [torch.empty((N, D), device="cuda") for _ in range(2)]
rew = torch.empty(N, device="cuda")
term = torch.empty(N, dtype=torch.bool, device="cuda")
trunc = torch.empty_like(term)
env.reset(out=obs[0])
lanes = {1: 1, 2: 2, 3: 4, 4: 4}[A]
nw = (E * lanes + 63) // 64
dbg = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
acts = torch.randint(0, 7, (8, N), device="cuda")
for i in range(int(os.environ.get("STEPS", "60"))):
    stamped_step(env, acts[i % 8], obs[i % 2], rew, term, trunc, dbg)
torch.cuda.synchronize()
full = dbg.cpu().numpy().reshape(nw, 16)
names = ["loads + table staging", "unpack", "step", "rewards + state store", "feature block", "block -> LDS image", "store stream"]
print(f"{layout} A={A} {feature} E={E}: {nw} waves, D={D}")
for k, name in enumerate(names):
    d = full[:, k + 1] - full[:, k]
    print(f"  {name:24s} cycles p50 {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}")
rt0, rt1, rt2 = full[:, 10], full[:, 11], full[:, 8]
base = rt0.min()
us = lambda x: (x - base) / 100.0  # noqa: E731
for name, x in (("wave start", rt0), ("image done", rt1), ("wave end", rt2)):
    print("  %-12s us: p0 %.2f p50 %.2f p90 %.2f p100 %.2f" % ((name,) + tuple(np.percentile(us(x), [0, 50, 90, 100]))))
h, _ = np.histogram(us(rt0), bins=np.arange(0, 60, 2.0))
print("  start-time histogram per 2 us:", h.tolist())
h, _ = np.histogram(us(rt2), bins=np.arange(0, 60, 2.0))
print("  end-time histogram per 2 us:  ", h.tolist())

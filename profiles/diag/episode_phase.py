"""us per launch in windows of the 1000-step episode (random actions: loose items pile up)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
E, A = 65536, 2
env = BatchedOvercookedVecEnv(get_layout("overcooked_cramped_room_v0"), A, "global_obs", E, seed=1234)
N, D = env.N, env.D
obs = [torch.empty((N, D), device="cuda") for _ in range(2)]
rew = torch.empty(N, device="cuda"); term = torch.empty(N, dtype=torch.bool, device="cuda"); trunc = torch.empty_like(term)
acts = torch.randint(0, 7, (32, N), device="cuda")
env.reset(out=obs[0])
out = []
for w in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100):
        env.step_into(acts[i % 32], obs[i % 2], rew, term, trunc)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) * 10)
print(os.environ.get("MARL_AMD_LIB", "default").split("libvar_")[-1], " ".join("%.2f" % x for x in out))

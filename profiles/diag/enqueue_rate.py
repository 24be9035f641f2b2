"""Host time to enqueue one env step (ctypes call, no synchronisation) against the launch's GPU time."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
E = 65536
env = BatchedOvercookedVecEnv(get_layout("overcooked_cramped_room_v0"), 2, "global_obs", E, seed=1)
N, D = env.N, env.D
obs = torch.empty((N, D), device="cuda"); rew = torch.empty(N, device="cuda")
term = torch.empty(N, dtype=torch.bool, device="cuda"); trunc = torch.empty_like(term)
env.reset(out=obs)
acts = torch.randint(0, 7, (8, N), device="cuda")
for i in range(64):
    env.step_into(acts[i % 8], obs, rew, term, trunc)
torch.cuda.synchronize()
K = 2048
t0 = time.perf_counter()
for i in range(K):
    env.step_into(acts[i % 8], obs, rew, term, trunc)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6*(t1-t0)/K:.2f} us per step, enqueue+drain {1e6*(t2-t0)/K:.2f} us per step")

"""Time per step when K consecutive random-action steps are worked off by one launch
(marl_env_step_many) against K single-step launches."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
E, A = int(os.environ.get("SM_E", 65536)), 2
env = BatchedOvercookedVecEnv(get_layout("overcooked_cramped_room_v0"), A, "global_obs", E, seed=1)
N, D = env.N, env.D
B = 842.0
for K in (1, 2, 4, 8, 16, 32):
    acts = torch.randint(0, 7, (K, N), device="cuda")
    obs = torch.empty((K, N, D), device="cuda"); rew = torch.empty((K, N), device="cuda")
    te = torch.empty((K, N), dtype=torch.bool, device="cuda"); tr = torch.empty_like(te)
    env.reset()
    reps = max(2, 256 // K)
    for _ in range(3):
        env.step_many_into(acts, obs, rew, te, tr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.step_many_into(acts, obs, rew, te, tr)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * K)
    print(f"K={K:3d}: {us:.2f} us per step, {N * B / us / 1e6:.2f} TB/s algorithmic, {N / us * 1e6:.3e} agent-steps/s")

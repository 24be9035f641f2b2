"""Multi-step launch (marl_env_step_many), rows kept: us per step for K steps per launch."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
E, A = int(os.environ.get("SM_E", 65536)), int(os.environ.get("SM_A", 2))
layout = os.environ.get("SM_LAYOUT", "overcooked_cramped_room_v0")
feat = os.environ.get("SM_FEAT", "global_obs")
env = BatchedOvercookedVecEnv(get_layout(layout), A, feat, E, seed=1)
N, D = env.N, env.D
B = 4 * D + 6 + 8 + 2 * 4 * env.state_words / A
out = []
for K in [int(k) for k in os.environ.get("SM_K", "32,128").split(",")]:
    acts = torch.randint(0, 7, (K, N), device="cuda")
    obs = torch.empty((K, N, D), device="cuda"); rew = torch.empty((K, N), device="cuda")
    te = torch.empty((K, N), dtype=torch.bool, device="cuda"); tr = torch.empty_like(te)
    env.reset()
    reps = max(4, 512 // K)
    for _ in range(2):
        env.step_many_into(acts, obs, rew, te, tr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.step_many_into(acts, obs, rew, te, tr)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * K)
    out.append(f"K={K}: {us:.2f} us/step {N * B / us / 1e6:.2f} TB/s")
print(f"XP={os.environ.get('MARL_AMD_XP', '0')} {layout} A={A} {feat} E={E}: " + " | ".join(out))

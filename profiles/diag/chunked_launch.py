"""Does a job beyond the last-level cache run faster as several launches of 65,536 envs than as
one launch?  Two (or more) env handles write into slices of one observation tensor."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
rows = get_layout("overcooked_counter_circuit_v0")
A = 2
def run(total, chunk, steps=300):
    envs = [BatchedOvercookedVecEnv(rows, A, "global_obs", chunk, seed=1, env_index_base=k * chunk) for k in range(total // chunk)]
    D = envs[0].D
    obs = [torch.empty((total * A, D), device="cuda") for _ in range(2)]
    rew = torch.empty(total * A, device="cuda"); term = torch.empty(total * A, dtype=torch.bool, device="cuda"); trunc = torch.empty_like(term)
    acts = torch.randint(0, 7, (8, total * A), device="cuda")
    n = chunk * A
    for k, e in enumerate(envs):
        e.reset(out=obs[0][k * n:(k + 1) * n])
    def step(i):
        for k, e in enumerate(envs):
            sl = slice(k * n, (k + 1) * n)
            e.step_into(acts[i % 8][sl], obs[i % 2][sl], rew[sl], term[sl], trunc[sl])
    for i in range(30):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        step(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    print(f"{total} envs as {total // chunk} launch(es) of {chunk}: {us:.1f} us per step, {total * A * D * 4 / us / 1e6:.2f} TB/s of rows")
for total in (131072, 262144, 1048576):
    for chunk in (total, 65536, 32768):
        run(total, chunk)

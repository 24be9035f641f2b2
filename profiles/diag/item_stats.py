"""How many loose items does a random-action cramped_room env carry (the item loop of the
feature block iterates over them; a wave pays for the largest count among its 32 envs)?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
E = 65536
env = BatchedOvercookedVecEnv(get_layout("overcooked_cramped_room_v0"), 2, "global_obs", E, seed=1234)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
for t in range(1, 2001):
    env.step(torch.randint(0, 7, (E * 2,), device="cuda", generator=g))
    if t in (100, 500, 999, 1500, 2000):
        items = env.export_state()["items"]
        n = (items != 0).sum(1)
        per_wave = n.reshape(-1, 32).max(1)
        print("t=%4d items per env: mean %.2f max %d | per-wave max: mean %.2f | hist %s" % (t, n.mean(), n.max(), per_wave.mean(), np.bincount(n, minlength=8)[:10].tolist()))

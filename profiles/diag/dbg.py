import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import oracle
from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
rows = get_layout("overcooked_cramped_room_v0")
E, A = 257, 2
gpu = BatchedOvercookedVecEnv(rows, A, "global_obs", E, max_steps=60, seed=1234, env_index_base=1000)
ora = oracle.OracleVecEnv(rows, A, "global_obs", max_steps=60, num_envs=E, seed=1234, env_index_base=1000)
gpu.reset(); ora.reset()
rng = np.random.default_rng(7)
for t in range(5):
    act = rng.choice(7, size=E*A, p=[0.14, 0.14, 0.14, 0.14, 0.3, 0.04, 0.1]).astype(np.int64)
    og, *_ = gpu.step(torch.from_numpy(act)); oo, *_ = ora.step(act)
    og = og[OBS_KEY].cpu().numpy()
    bad = np.nonzero((og != oo).any(axis=1))[0]
    for r in bad[:3]:
        cols = np.nonzero(og[r] != oo[r])[0]
        print(t, r, cols, og[r][cols], oo[r][cols])
    if bad.size: break

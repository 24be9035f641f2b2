"""Per-XCD end of the last wave over consecutive rows-kept launches: is the slow XCD always the same one?"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _stamped import stamped_step
E, A = int(os.environ.get("ST_E", 65536)), 2
RING = int(os.environ.get("ST_RING", 16))
env = BatchedOvercookedVecEnv(get_layout("overcooked_cramped_room_v0"), A, "global_obs", E, seed=1)
N, D = env.N, env.D
obs = torch.empty((RING, N, D), device="cuda"); rew = torch.empty(N, device="cuda")
term = torch.empty(N, dtype=torch.bool, device="cuda"); trunc = torch.empty_like(term)
env.reset(out=obs[0])
nw = N // 64
dbg = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
acts = torch.randint(0, 7, (8, N), device="cuda")
for i in range(2 * RING):
    stamped_step(env, acts[i % 8], obs[(i + 1) % RING], rew, term, trunc, dbg)
rows = []
for i in range(2 * RING, 2 * RING + 12):
    # a burst of launches, the stamps of the last one are read (launches in a row, as in a rollout)
    for j in range(4):
        stamped_step(env, acts[(i + j) % 8], obs[(4 * i + j + 1) % RING], rew, term, trunc, dbg)
    torch.cuda.synchronize()
    full = dbg.cpu().numpy().reshape(nw, 16)
    rt0, rt2, xcc = full[:, 10], full[:, 8], full[:, 9]
    base = rt0.min()
    ends = [(rt2[xcc == x].max() - base) / 100.0 for x in range(8)]
    rows.append(ends)
    print("launch %2d slot %2d: " % (i, (4 * i + 4) % RING) + " ".join("%5.2f" % e for e in ends) + "  | max %.2f mean %.2f" % (max(ends), np.mean(ends)))
rows = np.array(rows)
print("mean over launches: " + " ".join("%5.2f" % e for e in rows.mean(0)))

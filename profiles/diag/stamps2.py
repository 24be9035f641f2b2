import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _stamped import stamped_step
E, A = 65536, 2
env = BatchedOvercookedVecEnv(get_layout("overcooked_cramped_room_v0"), A, "global_obs", E, seed=1)
N, D = env.N, env.D
obs = [torch.empty((N, D), device="cuda") for _ in range(2)]; rew = torch.empty(N, device="cuda")
term = torch.empty(N, dtype=torch.bool, device="cuda"); trunc = torch.empty_like(term)
env.reset(out=obs[0])
nw = N // 64
dbg = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
acts = torch.randint(0, 7, (8, N), device="cuda")
for i in range(50):
    stamped_step(env, acts[i % 8], obs[i % 2], rew, term, trunc, dbg)
torch.cuda.synchronize()
full = dbg.cpu().numpy().reshape(nw, 16)
rt0, rt1, rt2 = full[:, 10], full[:, 11], full[:, 8]
base = rt0.min()
us = lambda x: (x - base) / 100.0
print("wave start  (us): p0 %.2f p50 %.2f p90 %.2f p100 %.2f" % tuple(np.percentile(us(rt0), [0, 50, 90, 100])))
print("sim done    (us): p0 %.2f p50 %.2f p90 %.2f p100 %.2f" % tuple(np.percentile(us(rt1), [0, 50, 90, 100])))
print("wave end    (us): p0 %.2f p50 %.2f p90 %.2f p100 %.2f" % tuple(np.percentile(us(rt2), [0, 50, 90, 100])))
print("sim dur     (us): p50 %.2f p90 %.2f" % tuple(np.percentile((rt1 - rt0) / 100.0, [50, 90])))
print("stream dur  (us): p10 %.2f p50 %.2f p90 %.2f" % tuple(np.percentile((rt2 - rt1) / 100.0, [10, 50, 90])))
hw = full[:, 12]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = full[:, 9]
key = xcc * 1000 + se * 100 + sh * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs used:", len(u), "waves per CU: min", cnt.min(), "max", cnt.max(), "hist", np.bincount(cnt))
# histogram of end times in 1us bins
h, _ = np.histogram(us(rt2), bins=np.arange(0, 30, 1.0))
print("end-time histogram per us:", h.tolist())
h, _ = np.histogram(us(rt0), bins=np.arange(0, 30, 1.0))
print("start-time histogram per us:", h.tolist())
# per-XCD view in wall-clock time
for x in range(8):
    m = xcc == x
    if m.any():
        print("xcc %d: start p50 %.2f  sim-done p50 %.2f  end p50 %.2f p100 %.2f  stream p50 %.2f" % (
            x, np.median(us(rt0[m])), np.median(us(rt1[m])), np.median(us(rt2[m])), us(rt2[m]).max(),
            np.median((rt2[m] - rt1[m]) / 100.0)))

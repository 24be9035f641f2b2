"""Where the waves of a rows-kept launch end: by XCD, by CU, by the order in which a CU's workgroups arrived."""
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
for i in range(3 * RING + 5):
    stamped_step(env, acts[i % 8], obs[(i + 1) % RING], rew, term, trunc, dbg)
torch.cuda.synchronize()
full = dbg.cpu().numpy().reshape(nw, 16)
rt0, rt1, rt2 = full[:, 10], full[:, 11], full[:, 8]
base = rt0.min()
us = lambda x: (x - base) / 100.0
print("ring", RING, "envs", E)
print("wave start  (us): p0 %.2f p50 %.2f p90 %.2f p100 %.2f" % tuple(np.percentile(us(rt0), [0, 50, 90, 100])))
print("sim done    (us): p0 %.2f p50 %.2f p90 %.2f p100 %.2f" % tuple(np.percentile(us(rt1), [0, 50, 90, 100])))
print("wave end    (us): p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f" % tuple(np.percentile(us(rt2), [0, 10, 50, 90, 100])))
h, _ = np.histogram(us(rt2), bins=np.arange(0, 30, 1.0))
print("end-time histogram per us:", h.tolist())
hw = full[:, 12]
slot = hw & 0xF; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = full[:, 9]
for x in range(8):
    m = xcc == x
    if m.any():
        print("xcc %d: waves %d  end p10 %.2f p50 %.2f p90 %.2f p100 %.2f" % ((x, m.sum()) + tuple(np.percentile(us(rt2[m]), [10, 50, 90, 100]))))
key = xcc * 4096 + se * 512 + sh * 256 + cu
u = np.unique(key)
last = np.array([us(rt2[key == k]).max() for k in u]); first = np.array([us(rt2[key == k]).min() for k in u])
print("per CU: last wave's end p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f" % tuple(np.percentile(last, [0, 10, 50, 90, 100])))
print("per CU: first wave's end p0 %.2f p50 %.2f p100 %.2f" % tuple(np.percentile(first, [0, 50, 100])))
print("wave slots seen:", np.bincount(slot).tolist(), " simd:", np.bincount(simd).tolist())
for s in np.unique(slot):
    m = slot == s
    print("slot %d: waves %d end p10 %.2f p50 %.2f p90 %.2f" % ((s, m.sum()) + tuple(np.percentile(us(rt2[m]), [10, 50, 90]))))
# by position in the grid (workgroup index = wave // 4)
wg = np.arange(nw) // 4
for lo in range(0, nw // 4, nw // 32):
    m = (wg >= lo) & (wg < lo + nw // 32)
    print("workgroups %4d..%4d: end p50 %.2f p100 %.2f" % (lo, lo + nw // 32 - 1, np.median(us(rt2[m])), us(rt2[m]).max()))

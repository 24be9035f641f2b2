"""-DOC_STAMPS build: when do the waves of a multi-step launch end, by XCD (last step's phase stamps, whole-launch wall clock)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from marl_amd import _lib
from marl_amd.env import BatchedOvercookedVecEnv, _ptr
from marl_amd.layouts import get_layout
E, A, K = 65536, 2, int(os.environ.get("ST_K", 32))
env = BatchedOvercookedVecEnv(get_layout("overcooked_cramped_room_v0"), A, "global_obs", E, seed=1)
N, D = env.N, env.D
obs = torch.empty((K, N, D), device="cuda"); rew = torch.empty((K, N), device="cuda")
term = torch.empty((K, N), dtype=torch.bool, device="cuda"); trunc = torch.empty_like(term)
env.reset()
nw = N // 64
dbg = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
acts = torch.randint(0, 7, (K, N), device="cuda")
def many():
    _lib.check(env._L.marl_env_step_many(env._h, _ptr(env.state), _ptr(acts), 8, K, _ptr(obs), _ptr(rew), _ptr(term), _ptr(trunc),
                                         _ptr(dbg), env._stream()), "marl_env_step_many")
for _ in range(3):
    many()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); many(); e1.record(); torch.cuda.synchronize()
print("launch of %d steps: %.1f us = %.2f us per step" % (K, e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / K))
full = dbg.cpu().numpy().reshape(nw, 16)
rt0, rt2, xcc, hw = full[:, 10], full[:, 8], full[:, 9], full[:, 12]
base = rt0.min()
us = lambda x: (x - base) / 100.0
print("wave start p0 %.2f p100 %.2f; wave end p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f" % (us(rt0).min(), us(rt0).max(), *np.percentile(us(rt2), [0, 10, 50, 90, 100])))
for x in range(8):
    m = xcc == x
    print("xcc %d: end p0 %.1f p50 %.1f p100 %.1f  (per step %.2f .. %.2f us)" % (x, us(rt2[m]).min(), np.median(us(rt2[m])), us(rt2[m]).max(), us(rt2[m]).min() / K, us(rt2[m]).max() / K))
slot = hw & 0xF
for s_ in np.unique(slot):
    m = slot == s_
    print("slot %d: end p10 %.1f p50 %.1f p90 %.1f" % (s_, *np.percentile(us(rt2[m]), [10, 50, 90])))
key = xcc * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF)
u = np.unique(key)
last = np.array([us(rt2[key == k]).max() for k in u]); first = np.array([us(rt2[key == k]).min() for k in u])
print("per CU (%d CUs): first wave's end p0 %.1f p50 %.1f p100 %.1f; last wave's end p0 %.1f p50 %.1f p100 %.1f" % (len(u), *np.percentile(first, [0, 50, 100]), *np.percentile(last, [0, 50, 100])))
# last step's phase stamps (cycles): 0 start of kernel .. 6 image done, 7 end
st = full[:, :8]
print("last step: sim+block+image cycles p50 %d; stream issue cycles p50 %d" % (np.median(st[:, 6] - st[:, 2]), np.median(st[:, 7] - st[:, 6])))

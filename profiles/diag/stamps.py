import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _stamped import stamped_step
E, A = int(os.environ.get("ST_E", 65536)), int(os.environ.get("ST_A", 2))
env = BatchedOvercookedVecEnv(get_layout(os.environ.get("ST_LAYOUT", "overcooked_cramped_room_v0")), A, os.environ.get("ST_FEAT", "global_obs"), E, seed=1)
N, D = env.N, env.D
obs = torch.empty((N, D), device="cuda"); rew = torch.empty(N, device="cuda")
term = torch.empty(N, dtype=torch.bool, device="cuda"); trunc = torch.empty_like(term)
env.reset(out=obs)
nw = (E + (64 // (1 if A == 1 else 2 if A == 2 else 4)) - 1) // (64 // (1 if A == 1 else 2 if A == 2 else 4))
dbg = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
acts = torch.randint(0, 7, (8, N), device="cuda")
for i in range(50):
    stamped_step(env, acts[i % 8], obs, rew, term, trunc, dbg)
torch.cuda.synchronize()
full = dbg.cpu().numpy().reshape(nw, 16)
s = full[:, :8]
t0 = s[:, 0].min()
d = np.diff(s, axis=1)
names = ["loads+stage+barrier", "unpack+key", "step_env", "rewards/store_state", "build_block", "write_block", "barrier+stream"]
print("per-wave cycles (median / p10 / p90):")
for k, n in enumerate(names):
    print(f"  {n:22s} {np.median(d[:,k]):8.0f} {np.percentile(d[:,k],10):8.0f} {np.percentile(d[:,k],90):8.0f}")
print("start spread (cycles):", np.percentile(s[:,0]-t0, [50, 90, 100]))
print("end - first start (cycles): median", np.median(s[:,7]-t0), "max", (s[:,7]-t0).max())
print("total per wave median", np.median(s[:,7]-s[:,0]))

rt = full[:, 8]
print("end realtime spread (100MHz ticks): min", rt.min()-rt.min(), "median", np.median(rt)-rt.min(), "max", rt.max()-rt.min())
for x in range(8):
    m = full[:, 9] == x
    if m.any():
        ss = s[m]
        print("xcc", x, "waves", m.sum(), "start spread", ss[:,0].max()-ss[:,0].min(), "end spread", ss[:,7].max()-ss[:,0].min(), "median life", np.median(ss[:,7]-ss[:,0]))

"""Wall-clock split of a training run between rollouts and updates (synchronising around updates)."""
import sys, time, torch
import marl_amd.agents.mappo as m
from marl_amd import train
acc = {"upd": 0.0, "n": 0}
orig = m.MAPPO.update
def upd(self, obs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orig(self, obs)
    torch.cuda.synchronize(); acc["upd"] += time.perf_counter() - t0; acc["n"] += 1
m.MAPPO.update = upd
t0 = time.perf_counter()
train.main(sys.argv[1:])
tot = time.perf_counter() - t0
print("updates %d: %.1f ms each; everything else %.1f ms per update cycle" % (acc["n"], acc["upd"] / acc["n"] * 1e3, (tot - acc["upd"]) / acc["n"] * 1e3))

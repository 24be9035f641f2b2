"""Where an update's wall time goes at the reference's batch size (16 envs, T=256, 5 epochs x 4
minibatches): rollout (graph-replayed) against GAE + PPO update, ms per update in steady state."""
import sys
import time
import types

import torch

sys.path.insert(0, ".")
from marl_amd import train  # noqa: E402
import marl_amd.agents.mappo as m  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t_update = []
orig = m.MAPPO.update


def upd(self, obs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    orig(self, obs)
    torch.cuda.synchronize()
    t_update.append(time.perf_counter() - t0)


m.MAPPO.update = upd
n = 300
t0 = time.perf_counter()
train.main(["--num-agents", "2", "--num-envs", str(E), "--layout", "overcooked_cramped_room_v0", "--num-steps", "256",
            "--num-minibatches", "4", "--total-steps", str(E * 2 * 256 * n), "--ppo-epoch", "5", "--clip-param", "0.05",
            "--value-loss-coef", "0.1", "--lr", "3e-4", "--feature", "global_obs", "--shuffle", "numpy", "--data-path", "/tmp/d"])
total = time.perf_counter() - t0
u = sum(t_update[50:]) / len(t_update[50:])
print(f"E={E}: update {u * 1e3:.2f} ms; whole run {total / n * 1e3:.2f} ms per update incl. start-up; "
      f"first 50 updates {sum(t_update[:50]) * 1e3:.0f} ms")

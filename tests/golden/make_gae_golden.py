"""Generates tests/golden/gae_golden.npz by running the REFERENCE's own compute_gae.

Run in the build container only (needs /root/reference on disk):

    python tests/golden/make_gae_golden.py

It imports agents/MAPPO.py and agents/CentralizedMAPPO.py from the reference tree and
calls their ``compute_gae`` (agents/MAPPO.py:82-114, agents/CentralizedMAPPO.py:48-82) and
the value-target line (agents/MAPPO.py:144) on seeded random inputs.  Those modules do
``from torch.utils.tensorboard import SummaryWriter`` at import time; tensorboard is not
installed here and is never touched by compute_gae, so an inert placeholder module is
registered for that one import.  Only inputs and outputs (data) are stored.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def main():
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["torch.utils.tensorboard"] = tb
    sys.path.insert(0, REF)
    from agents.MAPPO import MAPPO  # noqa: E402
    from agents.CentralizedMAPPO import CMAPPO  # noqa: E402

    torch.manual_seed(20250211)
    out = {}
    cases = [  # (T, E, A, gamma, lam, done_prob)
        (5, 3, 2, 0.99, 0.95, 0.3),
        (128, 16, 2, 0.99, 0.95, 0.01),
        (256, 16, 2, 0.99, 0.95, 0.004),
        (64, 8, 4, 0.9, 0.8, 0.1),
        (1, 4, 2, 0.99, 0.95, 0.5),
        (33, 67, 1, 0.97, 0.9, 0.05),
    ]
    for ci, (T, E, A, gamma, lam, dp) in enumerate(cases):
        N = E * A
        rewards = torch.where(torch.rand(T, N) < 0.2, torch.randn(T, N).abs(), torch.zeros(T, N))
        dones = (torch.rand(T, N) < dp).float()
        args = types.SimpleNamespace(num_envs=E)
        buf = types.SimpleNamespace(max_size=T)
        for mode, cls in ((0, MAPPO), (1, CMAPPO)):
            if mode == 0:
                values = torch.randn(T, N)
                next_values = torch.randn(1, N)
            else:
                # the centralised critic emits one value per step, broadcast over the rows
                # (agent_environment.py:83-84, buffer.py:28); bootstrap is (1, 1)
                values = torch.randn(T, 1).expand(T, N).contiguous()
                next_values = torch.randn(1, 1)
            agent = cls(None, None, None, buf, None, None, num_agents=A, gamma=gamma, lam=lam, args=args)
            agent.device = torch.device("cpu")
            adv = agent.compute_gae(rewards, dones, values, next_values)
            adv = adv.reshape(T, N)
            ret = adv + values  # agents/MAPPO.py:144
            k = f"c{ci}_m{mode}_"
            out[k + "rewards"] = rewards.numpy()
            out[k + "dones"] = dones.numpy()
            out[k + "values"] = values.numpy()
            out[k + "next_values"] = next_values.numpy().reshape(-1)
            out[k + "adv"] = adv.numpy()
            out[k + "ret"] = ret.numpy()
            out[k + "hp"] = np.array([gamma, lam], np.float64)
    out["num_cases"] = np.array(len(cases))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gae_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

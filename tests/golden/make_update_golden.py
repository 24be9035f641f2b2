"""Generates tests/golden/update_golden.npz by running the REFERENCE's own PPO update.

Run in the build container only (needs /root/reference on disk):

    python tests/golden/make_update_golden.py

It imports model.py, buffer.py, agents/MAPPO.py and agents/CentralizedMAPPO.py from the reference
tree (an inert placeholder stands in for tensorboard's SummaryWriter, which the update never
touches with log=False), fills a Buffer with seeded synthetic data, and calls ``update`` once
(agents/MAPPO.py:116-233) on the CPU.  Stored: the inputs, and of every parameter after the
update a strided sample plus its sum and absolute sum.  The initial weights are not stored: they
are seeded elementwise draws that the test repeats (checksums recorded here).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "update_golden.npz")
STRIDE = 41


def summary(sd):
    out = {}
    for k, v in sd.items():
        f = v.detach().reshape(-1).double()
        out[k + "/sample"] = v.detach().reshape(-1)[::STRIDE].numpy().copy()
        out[k + "/sums"] = np.array([f.sum().item(), f.abs().sum().item()])
    return out


def main():
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["torch.utils.tensorboard"] = tb
    sys.path.insert(0, REF)
    import builtins
    _print = builtins.print
    builtins.print = lambda *a, **k: None  # the reference's Buffer prints its shapes
    from agents.CentralizedMAPPO import CMAPPO  # noqa: E402
    from agents.MAPPO import MAPPO  # noqa: E402
    from buffer import Buffer  # noqa: E402
    from model import Agent  # noqa: E402

    T, E, A, D, K = 8, 4, 2, 16, 7
    N = E * A
    out = {"shape": np.array([T, E, A, D, K])}
    for name, cls in (("mappo", MAPPO), ("cmappo", CMAPPO)):
        torch.manual_seed(4242)
        obs_space = types.SimpleNamespace(shape=(D,))
        act_space = types.SimpleNamespace(n=K, shape=())
        net = Agent(obs_space, act_space, num_agents=A, num_envs=E)
        # the orthogonal initialisation goes through a QR factorisation whose last bits depend on
        # the host's LAPACK: overwrite it with seeded elementwise draws, which do not
        wg = torch.Generator().manual_seed(4242)
        with torch.no_grad():
            for p_ in net.parameters():
                p_.copy_(torch.randn(p_.shape, generator=wg) * (0.05 if p_.dim() > 1 else 0.01))
        out[name + "/init"] = np.array([sum(p.double().sum().item() for p in net.parameters()),
                                        sum(p.double().abs().sum().item() for p in net.parameters())])
        g = torch.Generator().manual_seed(99)
        buf = Buffer(D, A, E, max_size=T)
        buf.obs_buff = torch.randn(T, N, D, generator=g)
        buf.actions_buff = torch.randint(0, K, (T, N), generator=g).float()
        buf.rewards_buff = torch.where(torch.rand(T, N, generator=g) < 0.3, torch.rand(T, N, generator=g), torch.zeros(T, N))
        buf.dones_buff = (torch.rand(T, N, generator=g) < 0.1).float()
        with torch.no_grad():
            _, lp, _, v = net.get_action_and_value(buf.obs_buff.reshape(-1, D), buf.actions_buff.reshape(-1).long())
        buf.logprobs_buff = (lp.reshape(T, N) + 0.05 * torch.randn(T, N, generator=g)).contiguous()
        if name == "mappo":
            buf.values_buff = v.reshape(T, N).contiguous()
        else:  # one central value per step, broadcast over the rows (agent_environment.py:83-84, buffer.py:28)
            buf.values_buff = torch.randn(T, 1, generator=g).expand(T, N).contiguous()
        next_obs = torch.randn(N, D, generator=g)
        for key in ("obs_buff", "actions_buff", "rewards_buff", "dones_buff", "logprobs_buff", "values_buff"):
            out[f"{name}/{key}"] = getattr(buf, key).numpy().copy()
        out[f"{name}/next_obs"] = next_obs.numpy().copy()

        env = types.SimpleNamespace(single_observation_space=obs_space, single_action_space=act_space)
        args = types.SimpleNamespace(num_envs=E, num_steps=T, num_agents=A, centralised=(name == "cmappo"),
                                     layout="synthetic", seed=0)
        # plain SGD instead of main.py:294's Adam: Adam's first steps are lr * sign(gradient), which turns
        # float32 rounding noise on near-zero gradients into full-size differences; the optimizer is an
        # input of the learner, and what is pinned here is everything before optimizer.step()
        opt = torch.optim.SGD(net.parameters(), lr=0.05)
        agent = cls(env, opt, net, buf, (D,), K, batch_size=T * N, num_mini_batches=4, num_agents=A, ppo_epoch=3,
                    clip_param=0.2, value_loss_coef=0.5, entropy_coef=0.01, max_grad_norm=0.5, gamma=0.99, lam=0.95,
                    args=args)
        agent.device = torch.device("cpu")
        np.random.seed(777)
        agent.update(next_obs)
        for k, val in summary(net.state_dict()).items():
            out[f"{name}/after/{k}"] = val
    builtins.print = _print
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()

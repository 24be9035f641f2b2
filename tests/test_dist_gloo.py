"""world_size-2 checks of the multi-rank path on CPU (gloo): gradient averaging through the
flat all-reduce, and env sharding by global env index."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from marl_amd import dist_utils
    from marl_amd.policy import Agent
    from marl_amd.spaces import Box, Discrete

    r, lr, w = dist_utils.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist_utils.world_size() == world

    # --- gradient averaging
    torch.manual_seed(100 + rank)          # different weights per rank ...
    net = Agent(Box(-1, 1, (12,)), Discrete(7), num_agents=2, num_envs=2)
    dist_utils.broadcast_parameters(net)   # ... made identical
    ref = [p.detach().clone() for p in net.parameters()]
    gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
    dist.all_gather(gathered, ref[0])
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    x = torch.randn(5, 12)
    _, lp, _, v = net.get_action_and_value(x)
    (lp.sum() + v.sum()).backward()       # centralised head gets no gradient: exercised as "grad is None"
    local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    reducer = dist_utils.FlatGradAllReducer(net.parameters())
    reducer()
    assert any(g is None for g in local)
    for p, g in zip(net.parameters(), local):
        if g is None:
            assert p.grad is None  # parameters without a gradient take no part in the collective
            continue
        both = [torch.zeros_like(p) for _ in range(world)]
        dist.all_gather(both, g)
        torch.testing.assert_close(p.grad, sum(both) / world, rtol=1e-6, atol=1e-7)

    # --- env sharding: rank r simulates envs [r*E, (r+1)*E) of one 2E-env job
    import oracle
    from marl_amd.layouts import get_layout

    E, A, steps = 12, 2, 30
    rows = get_layout("overcooked_cramped_room_v0")
    shard = oracle.OracleVecEnv(rows, A, "global_obs", max_steps=11, num_envs=E, seed=5, env_index_base=rank * E)
    obs = [shard.reset()]
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 7, size=(steps, world * E * A))
    rews = []
    for t in range(steps):
        o, rwd, _, _ = shard.step(acts[t, rank * E * A:(rank + 1) * E * A])
        obs.append(o); rews.append(rwd)
    np.save(os.path.join(tmp, f"obs{rank}.npy"), np.stack(obs))
    np.save(os.path.join(tmp, f"rew{rank}.npy"), np.stack(rews))
    # a statistic reduced across ranks the way the rollout driver does it
    tot = torch.tensor([float(np.stack(rews).sum())], dtype=torch.float64)
    dist_utils.all_reduce_sum_(tot)
    dist.barrier()
    if rank == 0:
        whole = oracle.OracleVecEnv(rows, A, "global_obs", max_steps=11, num_envs=world * E, seed=5)
        wobs = [whole.reset()]
        wrew = []
        for t in range(steps):
            o, rwd, _, _ = whole.step(acts[t])
            wobs.append(o); wrew.append(rwd)
        wobs, wrew = np.stack(wobs), np.stack(wrew)
        got_obs = np.concatenate([np.load(os.path.join(tmp, f"obs{k}.npy")) for k in range(world)], axis=1)
        got_rew = np.concatenate([np.load(os.path.join(tmp, f"rew{k}.npy")) for k in range(world)], axis=1)
        np.testing.assert_array_equal(got_obs, wobs)
        np.testing.assert_array_equal(got_rew, wrew)
        assert abs(tot.item() - float(wrew.sum())) < 1e-9
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path, oracle_mod):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)

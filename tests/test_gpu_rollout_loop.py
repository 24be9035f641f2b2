"""The device-resident rollout driver (marl_amd/rollout.py) against the CPU restatement of the
reference's loop (oracle/rollout_loop.py, after agent_environment.py:34-112) run on the CPU oracle
env with the same actions: every stored row, the carried observation handed to update(), and the
episode statistics, across updates and episode ends -- including the reference's one-step-late
restart (an episode is max_steps + 1 driver steps; the row after a truncation holds the restarted
env's first observation with dones = 1)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _drive(oracle_mod, E, A, T, updates, max_steps, graph_rollout, fused_sampler, layout="overcooked_cramped_room_v0",
           feature="global_obs", seed=9, centralised=False, shards=1):
    from marl_amd.agents import CMAPPO, MAPPO
    from marl_amd.env import make_env, make_vector_env
    from marl_amd.layouts import get_layout
    from marl_amd.policy import Agent
    from marl_amd.rollout import agent_environment_loop
    from marl_amd.rollout_buffer import Buffer
    from oracle.rollout_loop import OracleDictEnv, ReplayAgent, reference_rollout_loop

    dev = torch.device("cuda:0")
    env = make_env(A, layout, feature, max_steps=max_steps)
    snaps = []

    class Recorder(CMAPPO if centralised else MAPPO):
        def update(self, next_obs):  # keep what was collected, learn nothing
            snap = {k: getattr(self.buffer, k + "_buff").clone() for k in
                    ("obs", "actions", "rewards", "dones", "logprobs", "values")}
            snap["next_obs"] = next_obs.clone()
            assert self.buffer.ptr == 0 and self.buffer.size == T  # a full rollout was stored
            snaps.append(snap)
            self.buffer.reset()

    vec = make_vector_env(E, env, shards=shards, seed=seed, device=dev)
    torch.manual_seed(0)
    net = Agent(vec.single_observation_space, vec.single_action_space, A, E).to(dev)
    if fused_sampler:
        net.use_fused_sampling(77)
    buf = Buffer(vec.D, A, E, max_size=T, device=dev)
    args = types.SimpleNamespace(num_steps=T, num_envs=E, num_agents=A, log=False, graph_rollout=graph_rollout)
    agent = Recorder(vec, None, net, buf, None, None, batch_size=E * A * T, num_agents=A, args=args)
    assert agent.can_act_into() == fused_sampler and agent.centralised == centralised
    torch.manual_seed(1)
    returns, freq = agent_environment_loop(agent, vec, dev, num_update=updates, args=args)
    assert len(snaps) == updates

    # the reference's loop, restated, on the CPU env with the recorded actions
    ora = oracle_mod.OracleVecEnv(get_layout(layout), A, feature, max_steps=max_steps, num_envs=E, seed=seed)
    ora.reset()  # make_vector_env issues one reset of its own (main.py:95) before the loop's
    replay = ReplayAgent([s["actions"].cpu().numpy().astype(np.int64) for s in snaps],
                         [s["logprobs"].cpu().numpy() for s in snaps], [s["values"].cpu().numpy() for s in snaps])
    log = []
    ref_returns, ref_freq = reference_rollout_loop(replay, OracleDictEnv(ora), updates, T, log=log)
    return snaps, replay.snaps, returns, freq, ref_returns, ref_freq, net, log


@pytest.mark.parametrize("fused_sampler", [False, True])
@pytest.mark.parametrize("graph_rollout", ["off", "on"])
def test_stored_rollouts_match_the_reference_loop(oracle_mod, graph_rollout, fused_sampler):
    E, A, T, updates, max_steps = 8, 2, 16, 5, 10
    snaps, ref, returns, freq, ref_returns, ref_freq, net, log = _drive(oracle_mod, E, A, T, updates, max_steps,
                                                                       graph_rollout, fused_sampler)
    assert [x[0] for x in log].count("reset") == 1 + (T * updates) // (max_steps + 1)
    for u, (got, want) in enumerate(zip(snaps, ref)):
        for k in ("obs", "actions", "rewards", "dones", "next_obs"):
            np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=f"update {u}: {k}")
        # log-probabilities and values are those of the policy on the stored observations (to float32
        # summation order: a 16-row rollout step and this 256-row call may run different GEMM kernels)
        with torch.no_grad():
            _, lp, _, v = net.get_action_and_value(got["obs"].reshape(-1, got["obs"].shape[-1]),
                                                   got["actions"].reshape(-1).long())
        torch.testing.assert_close(got["logprobs"].reshape(-1), lp, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(got["values"].reshape(-1), v.squeeze(1), rtol=1e-4, atol=1e-5)
    # 80 steps = 7 episodes of 11 driver steps; the row after each truncation carries dones = 1
    dones = np.concatenate([s["dones"][:, 0].cpu().numpy() for s in snaps])
    assert list(np.nonzero(dones)[0]) == [k for k in range(T * updates) if k % (max_steps + 1) == max_steps]
    assert len(returns) == len(ref_returns) == 7
    assert returns == ref_returns
    for k in ("frequency_delivery_per_episode", "frequency_plated_per_episode", "frequency_ingredient_in_pot_per_episode"):
        assert [float(x) for x in freq[k]] == [float(x) for x in ref_freq[k]], k


def test_four_agents_local_obs_and_uneven_lengths(oracle_mod):
    """Another layout, agent count and feature set; rollout length that is no power of two (the
    graph-replayed segments are cut into aligned power-of-two runs) and an episode length that
    divides nothing."""
    E, A, T, updates, max_steps = 5, 4, 12, 6, 7
    snaps, ref, returns, _, ref_returns, _, _, _ = _drive(oracle_mod, E, A, T, updates, max_steps, "on", True,
                                                         layout="overcooked_coordination_ring_v0", feature="local_obs",
                                                         seed=21)
    for u, (got, want) in enumerate(zip(snaps, ref)):
        for k in ("obs", "actions", "rewards", "dones", "next_obs"):
            np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=f"update {u}: {k}")
    assert returns == ref_returns and len(returns) == (T * updates) // (max_steps + 1)


@pytest.mark.parametrize("fused_sampler", [False, True])
def test_centralised_learner_stores_the_joint_value_in_every_row(oracle_mod, fused_sampler):
    """CMAPPO through the same driver (step-by-step store without the fused sampler, straight into
    the store with it): rows as the reference loop stores them, and as value of every row of a step
    the centralised critic's one number for the concatenated observations (agents/MAPPO.py:70-73,
    agent_environment.py:83-84)."""
    E, A, T, updates, max_steps = 4, 2, 8, 4, 6
    snaps, ref, returns, _, ref_returns, _, net, _ = _drive(oracle_mod, E, A, T, updates, max_steps, "on", fused_sampler,
                                                           layout="overcooked_forced_coordination_v0", centralised=True)
    for u, (got, want) in enumerate(zip(snaps, ref)):
        for k in ("obs", "actions", "rewards", "dones", "next_obs"):
            np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=f"update {u}: {k}")
        for t in range(T):
            with torch.no_grad():
                v = net.get_value(got["obs"][t], joint_obs=got["obs"][t].reshape(1, -1))
            torch.testing.assert_close(got["values"][t], v.view(1).expand(E * A), rtol=1e-4, atol=1e-5)
    assert returns == ref_returns and len(returns) == (T * updates) // (max_steps + 1)


@pytest.mark.parametrize("graph_rollout", ["off", "on"])
def test_rows_that_do_not_start_aligned_in_the_store(oracle_mod, graph_rollout):
    """Three agents and five envs: N * D = 15 * 309 floats, so the rows of step t do not start at
    a 16-byte boundary inside the rollout store and the driver stages observations through
    tensors of their own; what ends up stored is the same."""
    E, A, T, updates, max_steps = 5, 3, 9, 5, 8
    snaps, ref, returns, _, ref_returns, _, _, _ = _drive(oracle_mod, E, A, T, updates, max_steps, graph_rollout, True,
                                                         layout="overcooked_cramped_room_v0", feature="global_obs", seed=4)
    assert (E * A * snaps[0]["obs"].shape[-1]) % 4 != 0
    for u, (got, want) in enumerate(zip(snaps, ref)):
        for k in ("obs", "actions", "rewards", "dones", "next_obs"):
            np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=f"update {u}: {k}")
    assert returns == ref_returns and len(returns) == (T * updates) // (max_steps + 1)


def test_reference_shaped_loop_runs_on_the_host_array_env(oracle_mod):
    """Drop-in at the vec-env boundary: the restated reference loop (host arrays in and out, as
    agent_environment.py drives its env) runs unchanged on the numpy_io GPU env and stores what it
    stores on the CPU oracle env."""
    from marl_amd.env import make_env, make_vector_env
    from marl_amd.layouts import get_layout
    from oracle.rollout_loop import OracleDictEnv, ReplayAgent, reference_rollout_loop

    E, A, T, updates, max_steps, seed = 6, 2, 10, 4, 7, 13
    layout = "overcooked_forced_coordination_v0"
    rng = np.random.default_rng(1)
    acts = [rng.integers(0, 7, size=(T, E * A)) for _ in range(updates)]
    env = make_env(A, layout, "global_obs", max_steps=max_steps)
    vec = make_vector_env(E, env, numpy_io=True, seed=seed)
    got_agent, logs = ReplayAgent(acts), []
    got_returns, got_freq = reference_rollout_loop(got_agent, vec, updates, T, log=logs)
    ora = oracle_mod.OracleVecEnv(get_layout(layout), A, "global_obs", max_steps=max_steps, num_envs=E, seed=seed)
    ora.reset()  # make_vector_env's own reset (main.py:95)
    want_agent = ReplayAgent(acts)
    want_returns, want_freq = reference_rollout_loop(want_agent, OracleDictEnv(ora), updates, T)
    assert got_returns == want_returns and got_freq == want_freq
    for u, (got, want) in enumerate(zip(got_agent.snaps, want_agent.snaps)):
        for k in got:
            np.testing.assert_array_equal(got[k], want[k], err_msg=f"update {u}: {k}")
    assert [x[0] for x in logs].count("reset") == 1 + (T * updates) // (max_steps + 1)


@pytest.mark.parametrize("graph_rollout", ["off", "on"])
@pytest.mark.parametrize("shards,E,A,layout,feature", [
    (2, 8, 2, "overcooked_cramped_room_v0", "global_obs"),
    (4, 8, 4, "overcooked_coordination_ring_v0", "local_obs"),
])
def test_pipelined_shards_store_what_the_single_handle_driver_stores(oracle_mod, graph_rollout, shards, E, A, layout, feature):
    """The env as shards on streams of their own, every shard running act -> step for a whole run of
    steps without a join in between: held to the restated reference loop like the single-handle
    driver, and to the single-handle driver itself (same actions drawn, same rows stored)."""
    T, updates, max_steps = 16, 5, 10
    snaps, ref, returns, freq, ref_returns, ref_freq, net, _ = _drive(oracle_mod, E, A, T, updates, max_steps, graph_rollout,
                                                                     True, layout=layout, feature=feature, shards=shards)
    for u, (got, want) in enumerate(zip(snaps, ref)):
        for k in ("obs", "actions", "rewards", "dones", "next_obs"):
            np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=f"update {u}: {k}")
    assert returns == ref_returns and len(returns) == (T * updates) // (max_steps + 1)
    for k in ("frequency_delivery_per_episode", "frequency_plated_per_episode", "frequency_ingredient_in_pot_per_episode"):
        assert [float(x) for x in freq[k]] == [float(x) for x in ref_freq[k]], k
    one, *_ = _drive(oracle_mod, E, A, T, updates, max_steps, graph_rollout, True, layout=layout, feature=feature, shards=1)
    for u, (got, want) in enumerate(zip(snaps, one)):
        # a shard's GEMMs have fewer rows than the whole batch's and may sum in another order: the
        # policy outputs agree to float32 rounding, and the actions drawn from them here exactly
        for k in ("actions", "obs", "rewards", "dones", "next_obs"):
            assert torch.equal(got[k], want[k]), f"update {u}: {k} differs from the single-handle driver"
        torch.testing.assert_close(got["logprobs"], want["logprobs"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(got["values"], want["values"], rtol=1e-4, atol=1e-5)

"""ShardedVecEnv: a batch as K env handles on K streams reproduces the single-handle batch (and
the oracle) bit for bit, through the joined vec-env calls and through the pipelined form."""
import numpy as np
import pytest
import torch

from marl_amd.layouts import get_layout

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout,A,feature,E,K", [
    ("overcooked_cramped_room_v0", 2, "global_obs", 512, 2),
    ("overcooked_coordination_ring_v0", 4, "local_obs", 396, 4),
    ("overcooked_counter_circuit_v0", 2, "global_obs", 300, 3),
])
def test_joined_calls_equal_single_handle_and_oracle(oracle_mod, layout, A, feature, E, K):
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv
    from marl_amd.sharded_env import ShardedVecEnv

    rows = get_layout(layout)
    sh = ShardedVecEnv(rows, A, feature, E, shards=K, max_steps=25, seed=5, env_index_base=1000)
    one = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=25, seed=5, env_index_base=1000)
    ora = oracle_mod.OracleVecEnv(rows, A, feature, max_steps=25, num_envs=E, seed=5, env_index_base=1000)
    assert (sh.N, sh.D, sh.num_envs) == (one.N, one.D, one.num_envs)
    o_s, o_1, o_o = sh.reset()[0][OBS_KEY], one.reset()[0][OBS_KEY], ora.reset()
    assert torch.equal(o_s, o_1)
    np.testing.assert_array_equal(o_s.cpu().numpy(), o_o)
    rng = np.random.default_rng(K)
    for t in range(60):
        act = rng.integers(0, 7, size=E * A)
        o_s, r_s, te_s, tr_s, _ = sh.step(torch.from_numpy(act))
        o_1, r_1, te_1, tr_1, _ = one.step(torch.from_numpy(act))
        o_o, r_o, _, tr_o = ora.step(act)
        assert torch.equal(o_s[OBS_KEY], o_1[OBS_KEY]) and torch.equal(r_s, r_1) and torch.equal(tr_s, tr_1)
        np.testing.assert_array_equal(o_s[OBS_KEY].cpu().numpy(), o_o, err_msg=f"step {t}")
        np.testing.assert_array_equal(r_s.cpu().numpy(), r_o)
        np.testing.assert_array_equal(tr_s.cpu().numpy(), tr_o)
    snap_s, snap_1 = sh.export_state(), one.export_state()
    for k in snap_1:
        np.testing.assert_array_equal(snap_s[k], snap_1[k])
    assert torch.equal(sh.featurize(), o_1[OBS_KEY])


@pytest.mark.parametrize("captured", [False, True])
def test_pipelined_steps_equal_single_handle_steps(captured):
    """run_pipelined with actions known up front: every shard steps on its own stream without a join
    between steps; eagerly and replayed from one hipGraph (twice: the replay carries on from the
    state the first one left)."""
    from marl_amd.env import BatchedOvercookedVecEnv
    from marl_amd.sharded_env import ShardedVecEnv

    rows = get_layout("overcooked_cramped_room_v0")
    E, A, T = 1024, 2, 24
    sh = ShardedVecEnv(rows, A, "global_obs", E, shards=2, max_steps=9, seed=3)
    one = BatchedOvercookedVecEnv(rows, A, "global_obs", E, max_steps=9, seed=3)
    N, D, dev = sh.N, sh.D, sh.device
    g = torch.Generator(device="cpu").manual_seed(1)
    acts = torch.randint(0, 7, (T, N), generator=g).to(dev)
    obs = torch.zeros((T + 1, N, D), device=dev)
    rew = torch.zeros((T, N), device=dev)
    te = torch.zeros((T, N), dtype=torch.bool, device=dev)
    tr = torch.zeros_like(te)
    want_obs, want_rew, want_tr = torch.zeros_like(obs), torch.zeros_like(rew), torch.zeros_like(tr)
    sh.reset(out=obs[0]); one.reset(out=want_obs[0])

    def body(k, shard, r, step):
        shard.step_into(acts[step][r], obs[step + 1][r], rew[step][r], te[step][r], tr[step][r])

    rounds = 2 if captured else 1
    graph = None
    for rnd in range(rounds):
        if captured:
            if graph is None:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    sh.run_pipelined(range(T), body)
            graph.replay()
        else:
            sh.run_pipelined(range(T), body)
        for t in range(T):
            one.step_into(acts[t], want_obs[t + 1], want_rew[t], te[t].clone(), want_tr[t])
        assert torch.equal(obs[1:], want_obs[1:]) and torch.equal(rew, want_rew) and torch.equal(tr, want_tr)
        assert tr.any()
    for k, s in enumerate(sh.shards):
        assert torch.equal(s.state, one.state[:, k * (E // 2):(k + 1) * (E // 2)])


def test_shard_counts_that_cannot_work_are_refused():
    from marl_amd.sharded_env import ShardedVecEnv

    rows = get_layout("overcooked_cramped_room_v0")
    with pytest.raises(ValueError, match="multiple"):
        ShardedVecEnv(rows, 2, "global_obs", 7, shards=2)
    with pytest.raises(ValueError, match="16 bytes"):
        ShardedVecEnv(rows, 3, "global_obs", 6, shards=2)   # 3 envs x 3 agents x 309 floats per shard

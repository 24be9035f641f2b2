"""Pins the CPU oracle to everything the reference ships or can produce for this path."""
import os

import numpy as np

from marl_amd.layouts import get_layout

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_featurizer_reproduces_reference_observation(oracle_mod):
    """states/agent_0_dropping_onion_state.npy (used at utils.py:14-18): 2-agent cramped_room
    global_obs, agent 0 at (2,3) facing up with an onion, agent 1 at (3,3) facing down."""
    gold = np.load(os.path.join(GOLDEN, "agent_0_dropping_onion_state.npy"))
    assert gold.shape == (1, 202) and gold.dtype == np.float32
    env = oracle_mod.OracleVecEnv(get_layout("overcooked_cramped_room_v0"), 2, "global_obs", num_envs=1)
    env.reset()
    env.agents[0, 0] = (2, 3, 3, 1)
    env.agents[0, 1] = (3, 3, 1, 0)
    obs = env.featurize()
    np.testing.assert_array_equal(obs[0:1], gold)
    # agent 1's row is the same two blocks in the other order (overcooked_features.py:110-117)
    np.testing.assert_array_equal(obs[1, :101], gold[0, 101:])
    np.testing.assert_array_equal(obs[1, 101:], gold[0, :101])
    # local_obs is the own block only (overcooked_features.py:201,214)
    env_l = oracle_mod.OracleVecEnv(get_layout("overcooked_cramped_room_v0"), 2, "local_obs", num_envs=1)
    env_l.load_snapshot(env.snapshot())
    np.testing.assert_array_equal(env_l.featurize()[0], gold[0, :101])


def test_gae_reproduces_reference_outputs(oracle_mod):
    """tests/golden/gae_golden.npz holds outputs of the reference's own compute_gae
    (see make_gae_golden.py)."""
    g = np.load(os.path.join(GOLDEN, "gae_golden.npz"))
    for ci in range(int(g["num_cases"])):
        for mode in (0, 1):
            k = f"c{ci}_m{mode}_"
            gamma, lam = (float(x) for x in g[k + "hp"])
            adv, ret = oracle_mod.gae(g[k + "rewards"], g[k + "dones"], g[k + "values"], g[k + "next_values"],
                                      gamma, lam, mode)
            if mode == 0:
                np.testing.assert_array_equal(adv, g[k + "adv"], err_msg=k)
                np.testing.assert_array_equal(ret, g[k + "ret"], err_msg=k)
            else:  # the reward mean is summed in another order than torch's .mean()
                from gae_checks import check_centralised, exact_centralised

                ea, er = exact_centralised(g, k)
                print(k, "adv", check_centralised(adv, g[k + "adv"], ea, k + "adv"))
                print(k, "ret", check_centralised(ret, g[k + "ret"], er, k + "ret"))


def test_gae_last_row_uses_its_own_done_flag(oracle_mod):
    """agents/MAPPO.py:102-104: for t = T-1 the mask is 1 - dones[T-1]."""
    T, N = 3, 2
    r = np.ones((T, N), np.float32)
    v = np.zeros((T, N), np.float32)
    d = np.zeros((T, N), np.float32)
    d[T - 1, 0] = 1.0
    nv = np.full(N, 10.0, np.float32)
    adv, _ = oracle_mod.gae(r, d, v, nv, 0.5, 1.0, 0)
    assert adv[T - 1, 0] == 1.0          # bootstrap masked by the row's own flag
    assert adv[T - 1, 1] == 1.0 + 0.5 * 10.0
    assert adv[T - 2, 0] == 1.0          # dones[T-1] also cuts the recursion one row earlier
    assert adv[T - 2, 1] == 1.0 + 0.5 * adv[T - 1, 1]


def test_observation_widths_match_reference_checkpoints(oracle_mod):
    """models/policy_2_agents_seed_4.pth has a 512x202 first layer: 2-agent global_obs."""
    import json

    shapes = json.load(open(os.path.join(GOLDEN, "policy_2_agents_seed_4_shapes.json")))
    rows = get_layout("overcooked_cramped_room_v0")
    assert oracle_mod.OracleVecEnv(rows, 2, "global_obs").D == shapes["network.0.weight"][1] == 202
    assert oracle_mod.OracleVecEnv(rows, 2, "local_obs").D == 101  # overcooked_features.py:201
    assert oracle_mod.OracleVecEnv(rows, 4, "global_obs").D == 420
    assert oracle_mod.OracleVecEnv(rows, 4, "local_obs").D == 105
    assert oracle_mod.OracleVecEnv(rows, 2, "Minimal_spatial").D == 33
    assert oracle_mod.OracleVecEnv(rows, 2, "Minimal_spatial_other_agent_aware").D == 35
    assert oracle_mod.OracleVecEnv(rows, 2, "only_direction").D == 4

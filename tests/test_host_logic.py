"""CPU-side checks of the host mirror of the reference interface."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_layout_registry():
    from marl_amd import layouts

    names = layouts.layout_names()
    for n in ("overcooked_cramped_room_v0", "overcooked_forced_coordination_v0", "overcooked_counter_circuit_v0",
              "overcooked_coordination_ring_v0", "overcooked_asymmetric_advantages_v0", "large_overcooked_layout"):
        assert n in names
    large = layouts.get_layout("large_overcooked_layout")  # overcooked_config.py:48-60
    assert len(large) == 11 and len(large[0]) == 17 and "".join(large).count("U") == 4
    with pytest.raises(KeyError):
        layouts.get_layout("nope")
    with pytest.raises(ValueError):
        layouts.register_layout("ragged", ["###", "##"])
    with pytest.raises(ValueError):
        layouts.register_layout("weird", ["#?#"])


def test_env_spec_matches_what_the_reference_reads():
    from marl_amd.env import OBS_KEY, make_env, obs_dim

    env = make_env(num_agents=2, layout="overcooked_cramped_room_v0", feature="global_obs")
    assert env.observation_spaces[0][OBS_KEY].shape == (202,)   # main.py:206
    assert env.action_spaces[0].n == 7                           # main.py:207
    assert env.config["num_agents"] == 2                         # main.py:295
    assert env.config["max_steps"] == 1000                       # overcooked_config.py:108
    assert obs_dim("global_obs", 4) == 420 and obs_dim("local_obs", 4) == 105
    assert obs_dim("only_direction", 2) == 4
    with pytest.raises(KeyError):
        make_env(2, "overcooked_cramped_room_v0", "no_such_feature")


def test_vec_env_refuses_to_run_without_a_gpu():
    from marl_amd import _lib
    from marl_amd.env import BatchedOvercookedVecEnv

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.MarlAmdError):
        BatchedOvercookedVecEnv("overcooked_cramped_room_v0", 2, "global_obs", 4)


def test_policy_parameter_names_and_shapes_match_reference_checkpoint():
    """models/policy_2_agents_seed_4.pth (names and shapes recorded in tests/golden)."""
    from marl_amd.policy import Agent
    from marl_amd.spaces import Box, Discrete

    want = json.load(open(os.path.join(GOLDEN, "policy_2_agents_seed_4_shapes.json")))
    net = Agent(Box(-np.inf, np.inf, (202,)), Discrete(7), num_agents=2, num_envs=1)
    got = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert got == want
    x = torch.randn(6, 202)
    a, lp, ent, v = net.get_action_and_value(x)
    assert a.shape == (6,) and lp.shape == (6,) and ent.shape == (6,) and v.shape == (6, 1)
    assert net.get_value(x).shape == (6, 1)
    assert net.get_value(x[:2], joint_obs=x[:2].reshape(1, -1)).shape == (1, 1)
    with pytest.raises(ValueError):
        net.get_value(x, joint_obs=x.reshape(1, -1))
    # orthogonal init with gain sqrt(2), zero bias (model.py:6-9)
    w = net.network[2].weight.detach()
    torch.testing.assert_close(w @ w.T, 2.0 * torch.eye(256), atol=1e-4, rtol=0)
    assert net.actor.bias.abs().max() == 0


def test_buffer_semantics():
    from marl_amd.rollout_buffer import Buffer

    b = Buffer(5, 2, 3, max_size=4, device="cpu")
    assert b.obs_buff.shape == (4, 6, 5) and b.actions_buff.dtype == torch.float32  # buffer.py:11-16
    for t in range(5):
        b.add(torch.full((6, 5), float(t)), torch.full((6,), t), torch.zeros(6), torch.ones(6), torch.zeros(6),
              torch.full((6,), 2.0 * t))
    assert b.ptr == 1 and b.size == 4                      # ring pointer wrapped (buffer.py:30-31)
    assert b.obs_buff[0, 0, 0] == 4 and b.obs_buff[1, 0, 0] == 1
    slot = b.obs_slot()
    slot.fill_(7.0)
    b.add(slot, torch.zeros(6), torch.zeros(6), torch.zeros(6), torch.zeros(6), torch.zeros(6))
    assert (b.obs_buff[1] == 7).all()                      # zero-copy path leaves the row as written
    b.reset()
    assert b.ptr == 0 and b.size == 0


def test_snapshot_validation_rejects_bad_states():
    from marl_amd.env import BatchedOvercookedVecEnv
    from marl_amd.layouts import get_layout

    env = BatchedOvercookedVecEnv.__new__(BatchedOvercookedVecEnv)  # host-side checker only
    env.layout_rows = get_layout("overcooked_cramped_room_v0")
    env.H, env.W, env.A, env.NP, env.E = 6, 7, 2, 1, 1
    agents = np.array([[[2, 3, 3, 1], [3, 3, 1, 0]]], np.int32)
    pots = np.array([[[0, 30]]], np.int32)
    items = np.zeros((1, 42), np.uint8)
    misc = np.zeros((1, 2), np.int32)
    env._validate_snapshot(agents, pots, items, misc)
    bad = agents.copy(); bad[0, 0, :2] = (1, 3)             # on the pot
    with pytest.raises(ValueError):
        env._validate_snapshot(bad, pots, items, misc)
    bad = agents.copy(); bad[0, 1, :2] = (2, 3)             # two agents on one cell
    with pytest.raises(ValueError):
        env._validate_snapshot(bad, pots, items, misc)
    bad_items = items.copy(); bad_items[0, 1 * 7 + 3] = 1   # item inside the pot cell
    with pytest.raises(ValueError):
        env._validate_snapshot(agents, pots, bad_items, misc)
    bad_items = items.copy(); bad_items[0, 2 * 7 + 3] = 1   # item under an agent
    with pytest.raises(ValueError):
        env._validate_snapshot(agents, pots, bad_items, misc)
    env._closed = True


def test_train_cli_rejects_out_of_scope_algorithms():
    from marl_amd import train

    with pytest.raises(SystemExit):
        train.main(["--algorithm", "qmix"])
    args = train.build_parser().parse_args([])
    # defaults of main.py:107-147
    assert (args.num_agents, args.num_envs, args.num_steps, args.num_minibatches) == (4, 8, 128, 4)
    assert (args.gamma, args.lam, args.lr, args.ppo_epoch, args.clip_param) == (0.99, 0.95, 1e-4, 10, 0.2)
    assert (args.value_loss_coef, args.entropy_coef, args.max_grad_norm) == (0.5, 0.01, 0.5)


def test_every_environment_variable_the_code_reads_is_listed_in_the_readme():
    import glob
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py")]
    for pat in ("marl_amd/*.py", "marl_amd/agents/*.py", "marl_amd/csrc/*.hip", "marl_amd/csrc/*.inc", "marl_amd/csrc/*.h"):
        files += glob.glob(os.path.join(root, pat))
    used = set()
    for f in files:
        used |= set(re.findall(r"MARL_AMD_[A-Z_]+", open(f).read()))
    listed = set(re.findall(r"MARL_AMD_[A-Z_]+", open(os.path.join(root, "README.md")).read()))
    assert used and used <= listed, f"not in README.md: {sorted(used - listed)}"

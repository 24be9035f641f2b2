"""The committed replay recordings (tests/golden/replay_*.npz, written by
tests/golden/make_replay_fixture.py from this repository's CPU oracle: regression fixtures, not
vectors of the reference) against the oracle (CPU) and against the HIP env (GPU): two, three and
four agents, five layouts (the reference's large one among them), global, local and minimal
observations."""
import os

import numpy as np
import pytest

FIXTURES = [os.path.join(os.path.dirname(__file__), "golden", name + ".npz")
            for name in ("replay_cramped_room", "replay_ring_4_agents", "replay_cramped_3_agents", "replay_forced_2_agents",
                         "replay_large_4_agents", "replay_circuit_minimal")]


@pytest.mark.parametrize("fixture", FIXTURES, ids=[os.path.basename(f)[:-4] for f in FIXTURES])
def test_oracle_reproduces_the_recording(oracle_mod, fixture):
    with np.load(fixture, allow_pickle=False) as z:
        rec = {k: z[k] for k in z.files}
    rows = [str(r) for r in rec["layout"]]
    W = len(rows[0])
    K, E, A = rec["actions"].shape
    ora = oracle_mod.OracleVecEnv(rows, A, str(rec["feature"]), max_steps=int(rec["max_steps"]), num_envs=E, seed=123,
                                  auto_reset=False)  # the seed must not matter: every random choice is given
    cells = rec["spawn_cells"].astype(np.int64)
    obs = ora.reset(spawn_override=(cells[..., 0] * W + cells[..., 1]).astype(np.uint8).reshape(-1),
                    dir_override=rec["spawn_dirs"].astype(np.uint8).reshape(-1))
    np.testing.assert_array_equal(obs.reshape(E, A, -1), rec["obs"][0])
    for k in range(K):
        obs, r, _, tr = ora.step(rec["actions"][k].reshape(-1).astype(np.int64),
                                 prio_override=rec["serve_rank"][k].reshape(-1))
        np.testing.assert_array_equal(obs.reshape(E, A, -1), rec["obs"][k + 1], err_msg=f"step {k}")
        np.testing.assert_array_equal(r.reshape(E, A), rec["rewards"][k])
        np.testing.assert_array_equal(tr.reshape(E, A), rec["truncated"][k])
        np.testing.assert_array_equal(ora.agents, rec["agents"][k + 1])
        np.testing.assert_array_equal(ora.pots, rec["pots"][k + 1])
        np.testing.assert_array_equal(ora.items, rec["items"][k + 1])
    assert rec["rewards"].sum() > 0 and rec["truncated"][-1].all() and not rec["truncated"][:-1].any()


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", FIXTURES, ids=[os.path.basename(f)[:-4] for f in FIXTURES])
def test_kernel_replays_the_recording(fixture):
    from marl_amd import replay

    report = replay.replay(fixture)
    compared = sorted(k for k in report if k != "produced")
    assert compared == ["agents", "items", "obs", "pots", "rewards", "truncated"]
    assert all(report[k] is None for k in compared), {k: report[k] for k in compared if report[k]}

"""tools/record_cogrid_episode.py without cogrid: the recorder reads a stand-in env with cogrid's
attribute surface (tests/cogrid_stub.py, dynamics = the CPU oracle), its output passes the replay
loader's schema checks and replays exactly -- through the oracle on the CPU, through the HIP env
on the GPU -- with the serving order it reconstructed from the outcomes."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CASES = [("overcooked_cramped_room_v0", 2, "global_obs", 60), ("overcooked_forced_coordination_v0", 2, "local_obs", 40),
         ("overcooked_coordination_ring_v0", 2, "Minimal_spatial_other_agent_aware", 50)]


def _record(oracle_mod, layout, A, feature, max_steps, envs=6):
    import record_cogrid_episode as rc
    from cogrid_stub import StubOvercooked
    from marl_amd.layouts import get_layout

    rows = get_layout(layout)
    seeds = iter(range(100, 100 + envs))
    return rc.record(lambda: StubOvercooked(oracle_mod, rows, A, feature, max_steps, next(seeds)), envs, max_steps + 5,
                     seed=3, feature=feature, max_steps=max_steps), rows


@pytest.mark.parametrize("layout,A,feature,max_steps", CASES)
def test_recording_has_the_replay_schema_and_replays_on_the_oracle(oracle_mod, tmp_path, layout, A, feature, max_steps):
    from marl_amd.replay import load_recording

    rec, rows = _record(oracle_mod, layout, A, feature, max_steps)
    path = tmp_path / "rec.npz"
    np.savez_compressed(path, **rec)
    got = load_recording(str(path))              # as `python -m marl_amd.replay rec.npz` reads it
    assert got["rows"] == [r.replace("+", " ") for r in rows]   # the grid as seen through env.grid.get
    K, E = got["actions"].shape[:2]
    assert K == max_steps and got["truncated"][-1].all() and got["contested"].any()
    assert {"obs", "agents", "items", "pots", "rewards", "serve_rank"} <= set(got)
    W = len(rows[0])
    ora = oracle_mod.OracleVecEnv(got["rows"], A, feature, max_steps=max_steps, num_envs=E, seed=999, auto_reset=False)
    cells = got["spawn_cells"].astype(np.int64)
    obs = ora.reset(spawn_override=(cells[..., 0] * W + cells[..., 1]).astype(np.uint8).reshape(-1),
                    dir_override=got["spawn_dirs"].astype(np.uint8).reshape(-1))
    np.testing.assert_array_equal(obs.reshape(E, A, -1), got["obs"][0])
    for k in range(K):
        obs, r, _, tr = ora.step(got["actions"][k].reshape(-1), prio_override=got["serve_rank"][k].reshape(-1))
        np.testing.assert_array_equal(obs.reshape(E, A, -1), got["obs"][k + 1], err_msg=f"step {k}")
        np.testing.assert_array_equal(r.reshape(E, A), got["rewards"][k])
        np.testing.assert_array_equal(ora.agents, got["agents"][k + 1])
        np.testing.assert_array_equal(ora.pots, got["pots"][k + 1])
        np.testing.assert_array_equal(ora.items, got["items"][k + 1])


def test_loader_rejects_malformed_recordings(oracle_mod):
    from marl_amd.replay import load_recording

    rec, _ = _record(oracle_mod, *CASES[0], envs=2)
    for name, bad in (("actions", rec["actions"][:, :1]), ("agents", rec["agents"][1:]), ("obs", rec["obs"][:, :, :1]),
                      ("spawn_cells", rec["spawn_cells"] + 100)):
        with pytest.raises(ValueError):
            load_recording(dict(rec, **{name: bad}))
    with pytest.raises(ValueError):
        load_recording(dict(rec, max_steps=3))


@pytest.mark.gpu
@pytest.mark.parametrize("layout,A,feature,max_steps", CASES)
def test_recording_replays_on_the_gpu(oracle_mod, layout, A, feature, max_steps):
    from marl_amd import replay

    rec, _ = _record(oracle_mod, layout, A, feature, max_steps)
    report = replay.replay(rec)
    compared = sorted(k for k in report if k != "produced")
    assert compared == ["agents", "items", "obs", "pots", "rewards", "truncated"]
    assert all(report[k] is None for k in compared), {k: report[k] for k in compared if report[k]}
    # a tampered recording is reported with the step, the feature and whether a contested step came first
    rec["obs"] = rec["obs"].copy()
    rec["obs"][5, 1, 0, 0] += 1
    assert "contested step" in replay.replay(rec)["obs"]

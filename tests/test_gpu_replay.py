"""The replay harness (marl_amd.replay) on a recording made with the CPU oracle standing in for
the env whose episode was recorded: explicit spawn cells, directions and serving ranks."""
import numpy as np
import pytest

from marl_amd.layouts import get_layout

pytestmark = pytest.mark.gpu


def _record(oracle_mod, layout, A, feature, E, K, with_state):
    rows = get_layout(layout)
    H, W = len(rows), len(rows[0])
    rng = np.random.default_rng(E + K)
    floor = [(r, c) for r in range(H) for c in range(W) if rows[r][c] in " +"]
    cells = np.stack([np.array(floor)[rng.permutation(len(floor))[:A]] for _ in range(E)])  # [E, A, 2]
    dirs = rng.integers(0, 4, size=(E, A))
    ora = oracle_mod.OracleVecEnv(rows, A, feature, max_steps=K + 5, num_envs=E, seed=0, auto_reset=False)
    obs = [ora.reset(spawn_override=(cells[..., 0] * W + cells[..., 1]).astype(np.uint8).reshape(-1),
                     dir_override=dirs.astype(np.uint8).reshape(-1))]
    snaps = [(ora.agents.copy(), ora.pots.copy(), ora.items.copy())]
    acts = rng.choice(7, size=(K, E, A), p=[0.15, 0.15, 0.15, 0.15, 0.3, 0.03, 0.07])
    rank = rng.integers(0, 200, size=(K, E, A))
    rew, trunc = [], []
    for k in range(K):
        o, r, _, tr = ora.step(acts[k].reshape(-1).astype(np.int64), prio_override=rank[k].reshape(-1).astype(np.uint8))
        obs.append(o); rew.append(r); trunc.append(tr)
        snaps.append((ora.agents.copy(), ora.pots.copy(), ora.items.copy()))
    D = obs[0].shape[1]
    rec = {"layout": np.array(rows), "feature": np.array(feature), "max_steps": np.array(K + 5), "spawn_cells": cells,
           "spawn_dirs": dirs, "actions": acts, "serve_rank": rank, "obs": np.stack(obs).reshape(K + 1, E, A, D),
           "rewards": np.stack(rew).reshape(K, E, A), "truncated": np.stack(trunc).reshape(K, E, A)}
    if with_state:
        rec["agents"] = np.stack([s[0] for s in snaps]); rec["pots"] = np.stack([s[1] for s in snaps])
        rec["items"] = np.stack([s[2] for s in snaps])
    return rec


@pytest.mark.parametrize("layout,A,feature,with_state", [
    ("overcooked_cramped_room_v0", 2, "global_obs", False), ("overcooked_cramped_room_v0", 2, "global_obs", True),
    ("overcooked_coordination_ring_v0", 4, "local_obs", True), ("overcooked_forced_coordination_v0", 2, "Minimal_spatial", False),
])
def test_recording_replays_identically(oracle_mod, tmp_path, layout, A, feature, with_state):
    from marl_amd import replay

    rec = _record(oracle_mod, layout, A, feature, E=40, K=120, with_state=with_state)
    assert rec["rewards"].sum() > 0 or feature == "Minimal_spatial"
    path = tmp_path / "episode.npz"
    np.savez(path, **rec)
    report = replay.replay(str(path))
    compared = [k for k in report if k != "produced"]
    assert set(compared) == {"obs", "rewards", "truncated"} | ({"agents", "pots", "items"} if with_state else set())
    assert all(report[k] is None for k in compared), {k: report[k] for k in compared if report[k]}
    assert replay.main([str(path)]) == 0


def test_a_difference_is_named(oracle_mod, tmp_path):
    from marl_amd import replay

    rec = _record(oracle_mod, "overcooked_cramped_room_v0", 2, "global_obs", E=8, K=20, with_state=False)
    rec["obs"][5, 3, 1, 101 + 4] += 1.0  # the other agent's inventory, first entry
    rec["rewards"][7, 2, 0] += 0.5
    report = replay.replay(rec)
    assert "after step 4, env 3, agent 1, position 105 (inventory[0] in the block of other agent #1)" in report["obs"]
    assert report["rewards"].startswith("1 entries differ") and report["truncated"] is None
    path = tmp_path / "bad.npz"
    np.savez(path, **rec)
    assert replay.main([str(path)]) == 1
    assert replay.feature_at("global_obs", 2, 95) == "agent_position[0] in the own block"
    assert replay.feature_at("local_obs", 4, 104) == "can_move_direction[3] in the own block"

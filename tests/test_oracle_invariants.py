"""Invariants of the Overcooked dynamics under random play, checked on the CPU oracle for every
registered layout with one to four agents (the GPU suite holds the HIP env to the oracle bit for
bit, so they hold for it too): agents stay on floor cells and never share one, every code stays
in its range, loose items only lie on floor or counters, objects are neither created from nothing
nor lost (an onion leaves the world only inside a delivered soup, a plate only under one), rewards
are sums of the configured amounts, and an episode truncates exactly at max_steps."""
import numpy as np
import pytest

from marl_amd.layouts import get_layout, layout_names

NONE, ONION, PLATE, SOUP = range(4)
COOK_TIME = 30


def _biased(rng, n):
    return rng.choice(7, size=n, p=[0.14, 0.14, 0.14, 0.14, 0.3, 0.04, 0.1]).astype(np.int64)


@pytest.mark.parametrize("layout", layout_names())
@pytest.mark.parametrize("A", [1, 2, 3, 4])
def test_invariants_under_random_play(oracle_mod, layout, A):
    rows = get_layout(layout)
    H, W = len(rows), len(rows[0])
    flat = "".join(rows)
    floor = np.array([ch in " +" for ch in flat])
    spawns = flat.count("+")
    if floor.sum() < A or 0 < spawns < A:
        # no start cell for every agent: refused at creation (the HIP library refuses the same layouts,
        # tests/native/capi_driver.cpp and tests/test_gpu_env_parity.py)
        with pytest.raises(ValueError):
            oracle_mod.OracleVecEnv(rows, A, "global_obs", max_steps=10, num_envs=2, seed=1)
        return
    holds_item = np.array([ch in " +C" for ch in flat])
    E, max_steps, steps = 48, 90, 200
    env = oracle_mod.OracleVecEnv(rows, A, "global_obs", max_steps=max_steps, num_envs=E, seed=17)
    env.reset()
    rng = np.random.default_rng(3)
    allowed = {round(0.1 * a + 0.3 * b + 1.0 * c, 4) for a in range(A + 1) for b in range(A + 1) for c in range(A + 1)}
    t_expected = 0
    for k in range(steps):
        before_soups = _count(env, SOUP)
        _, rew, term, trunc = env.step(_biased(rng, E * A))
        t_expected += 1
        assert not term.any()
        if t_expected == max_steps:
            assert trunc.all(), f"step {k}: every env truncates at max_steps"
            t_expected = 0  # the vector env restarted them
        else:
            assert not trunc.any(), f"step {k}: truncation before max_steps"
        assert (env.misc[:, 0] == t_expected).all()
        ag = env.agents
        cell = ag[..., 0] * W + ag[..., 1]
        assert ((ag[..., 0] >= 0) & (ag[..., 0] < H) & (ag[..., 1] >= 0) & (ag[..., 1] < W)).all()
        assert floor[cell].all(), f"step {k}: an agent left the floor"
        if A > 1:
            srt = np.sort(cell, axis=1)
            assert (srt[:, 1:] != srt[:, :-1]).all(), f"step {k}: two agents share a cell"
        assert ((ag[..., 2] >= 0) & (ag[..., 2] < 4)).all() and ((ag[..., 3] >= 0) & (ag[..., 3] <= SOUP)).all()
        assert (env.items <= SOUP).all()
        assert not env.items[:, ~holds_item].any(), f"step {k}: an item lies on a cell that cannot hold one"
        if env.NP:
            cnt, timer = env.pots[..., 0], env.pots[..., 1]
            assert ((cnt >= 0) & (cnt <= 3)).all() and ((timer >= 0) & (timer <= COOK_TIME)).all()
            assert (timer[cnt < 3] == 0).all() or (timer[cnt < 3] == COOK_TIME).all(), "only a full pot cooks"
        assert all(round(float(r), 4) in allowed for r in np.unique(rew)), np.unique(rew)
        if t_expected != 0:
            # soups appear only by plating a cooked pot (0.3 to the one who plates) and vanish only by delivery (1.0)
            per_env = rew.reshape(E, A)
            delivered = (per_env >= 1.0 - 1e-6).any(axis=1)
            after = _count(env, SOUP)
            assert (after[~delivered] >= before_soups[~delivered]).all(), f"step {k}: a soup vanished without a delivery"


def _count(env, kind):
    return (env.agents[..., 3] == kind).sum(axis=1) + (env.items == kind).sum(axis=1)

"""marl_reward_events (through EpisodeStats) vs the numpy restatement, exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# every value one agent-step can earn: own shaped reward in float64 + deliveries, then float32
_POSSIBLE = np.array([float(np.float32(own + k)) for own in (0.0, 0.1, 0.3) for k in range(0, 5)], dtype=np.float32)


@pytest.mark.parametrize("E,A", [(1, 1), (1, 2), (3, 4), (257, 2), (5000, 4), (65536, 2), (300001, 1)])
def test_totals_match_numpy(oracle_mod, E, A):
    from marl_amd.episode_stats import EpisodeStats

    rng = np.random.default_rng(E * 7 + A)
    stats = EpisodeStats(E, A, "cuda")
    totals = np.zeros(4, dtype=np.int64)
    first = np.zeros(4, dtype=np.float64)
    for step in range(5):
        p = np.full(_POSSIBLE.size, 0.02)
        p[0] = 1.0 - 0.02 * (_POSSIBLE.size - 1)
        r = rng.choice(_POSSIBLE, size=E * A, p=p).astype(np.float32)
        if step == 1:
            r[:] = 0.0
        stats.add(torch.from_numpy(r).cuda())
        t, f = oracle_mod.reward_events(r, E, A)
        totals += t
        first += f
    got_t, got_f = stats.raw()
    np.testing.assert_array_equal(got_t.cpu().numpy(), totals)
    np.testing.assert_array_equal(got_f.cpu().numpy(), first)
    d = stats.read()
    assert d["all_deliveries"] == totals[1] / E and d["first_rows_return"] == first[0]
    stats.reset()
    assert int(stats.raw()[0].abs().sum()) == 0 and stats.steps == 0


def test_rejects_bad_rewards():
    from marl_amd.episode_stats import EpisodeStats

    stats = EpisodeStats(4, 2, "cuda")
    with pytest.raises(ValueError):
        stats.add(torch.zeros(8, dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        stats.add(torch.zeros(7, dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):
        stats.add(torch.zeros(8, dtype=torch.float32))


@pytest.mark.parametrize("E,A,K", [(1, 2, 1), (16, 2, 7), (257, 4, 5), (5000, 2, 33)])
def test_many_steps_at_once_equal_single_steps(oracle_mod, E, A, K):
    """marl_reward_events_many over K stored steps == K marl_reward_events calls in order, exactly
    (integer totals; the rows-0..1 return is a double accumulated step after step)."""
    from marl_amd.episode_stats import EpisodeStats

    rng = np.random.default_rng(E + K)
    p = np.full(_POSSIBLE.size, 0.03)
    p[0] = 1.0 - 0.03 * (_POSSIBLE.size - 1)
    r = rng.choice(_POSSIBLE, size=(K, E * A), p=p).astype(np.float32)
    rd = torch.from_numpy(r).cuda()
    one, many = EpisodeStats(E, A, "cuda"), EpisodeStats(E, A, "cuda")
    for k in range(K):
        one.add(rd[k])
    many.add_many(rd)
    many.add_many(rd[:0])
    assert torch.equal(one.raw()[0], many.raw()[0]) and torch.equal(one.raw()[1], many.raw()[1])
    assert one.steps == many.steps == K
    totals = np.zeros(4, np.int64); first = np.zeros(4, np.float64)
    for k in range(K):
        t, f = oracle_mod.reward_events(r[k], E, A)
        totals += t; first += f
    np.testing.assert_array_equal(many.raw()[0].cpu().numpy(), totals)
    np.testing.assert_array_equal(many.raw()[1].cpu().numpy(), first)

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

"""marl_sample_categorical vs its numpy restatement and vs torch.distributions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,K", [(1, 7), (257, 7), (100000, 7), (5000, 3), (300, 1), (4096, 33)])
def test_matches_restatement_and_torch(oracle_mod, N, K):
    from marl_amd.sampling import CategoricalSampler

    rng = np.random.default_rng(N + K)
    logits = (rng.normal(size=(N, K)) * 3).astype(np.float32)
    logits[::7] *= 5  # some peaked rows: |logit| up to ~60, where one float32 ulp is 4e-6 (the tolerance below)
    t = torch.from_numpy(logits).cuda()
    s = CategoricalSampler(seed=99, row_index_base=1000)
    for call in range(3):
        a, lp, ent = s.sample(t)
        a_o, lp_o, ent_o = oracle_mod.sample_categorical(logits, 99, call, 1000)
        a = a.cpu().numpy()
        # exp differs in the last bit between numpy and the device: a draw that lands within
        # that of a bin edge may fall on the other side
        same = a == a_o
        assert same.mean() >= 1 - 1e-4, f"{(~same).sum()} of {N} actions differ"
        np.testing.assert_allclose(lp.cpu().numpy()[same], lp_o[same], rtol=3e-6, atol=1e-5)
        np.testing.assert_allclose(ent.cpu().numpy(), ent_o, rtol=3e-6, atol=1e-5)
        d = torch.distributions.Categorical(logits=t)
        torch.testing.assert_close(lp, d.log_prob(torch.from_numpy(a).cuda()), rtol=3e-6, atol=1e-5)
        torch.testing.assert_close(ent, d.entropy(), rtol=3e-6, atol=1e-5)
    assert int(s.counter) == 3


def test_frequencies_follow_the_distribution():
    from marl_amd.sampling import CategoricalSampler

    probs = torch.tensor([0.5, 0.02, 0.2, 0.08, 0.1, 0.07, 0.03])
    n = 400000
    t = probs.log().repeat(n, 1).cuda()
    a, _, _ = CategoricalSampler(seed=1).sample(t, want_entropy=False)
    freq = torch.bincount(a, minlength=7).float().cpu() / n
    assert (freq - probs).abs().max() < 4e-3
    # shards draw what the whole job draws; another call draws something else
    b, _, _ = CategoricalSampler(seed=1, row_index_base=n // 2).sample(t[n // 2:], want_entropy=False)
    assert torch.equal(a[n // 2:], b)
    s = CategoricalSampler(seed=1)
    first, _, _ = s.sample(t[:1000])
    second, _, _ = s.sample(t[:1000])
    assert not torch.equal(first, second)


def test_wide_rows_slots_and_caller_buffers():
    """Rows wider than the action count (the critic's value riding in the next column), results
    written into caller tensors, and per-call counter slots: same draws as the plain call."""
    from marl_amd.sampling import CategoricalSampler

    torch.manual_seed(3)
    n, k = 3000, 7
    wide = torch.randn(n, k + 1, device="cuda") * 2
    plain = CategoricalSampler(seed=5, row_index_base=40)
    slotted = CategoricalSampler(seed=5, row_index_base=40)
    slotted.use_slots(4)
    for rnd in range(2):
        for call in range(4):
            a0, lp0, _ = plain.sample(wide[:, :k].contiguous())
            a1 = torch.empty(n, dtype=torch.int64, device="cuda")
            lp1 = torch.empty(n, device="cuda")
            v1 = torch.empty(n, device="cuda")
            out = slotted.sample(wide, want_entropy=False, actions_out=a1, logprob_out=lp1, slot=call, num_actions=k,
                                 next_column_out=v1)
            assert out[0] is a1 and out[1] is lp1 and out[2] is None
            assert torch.equal(a0, a1) and torch.equal(lp0, lp1) and torch.equal(v1, wide[:, k])
        slotted.advance_round()
    assert int(plain.counter) == int(slotted.counter) == 8
    with pytest.raises(ValueError):
        slotted.sample(wide[:, :k].contiguous(), next_column_out=torch.empty(n, device="cuda"))
    with pytest.raises(ValueError):
        slotted.sample(wide, actions_out=torch.empty(n, dtype=torch.int32, device="cuda"))

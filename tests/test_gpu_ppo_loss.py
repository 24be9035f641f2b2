"""marl_ppo_loss against the autograd evaluation of the reference's formulas
(agents/MAPPO.py:167-218, model.py:52-56).  Floating point: tolerances stated per check."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(logits, values, actions, old_lp, adv, target, clip, ec, vc):
    dist = torch.distributions.Categorical(logits=logits)
    newlp = dist.log_prob(actions)
    entropy = dist.entropy()
    logratio = newlp - old_lp
    ratio = logratio.exp()
    pg = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
    vl = 0.5 * ((values.squeeze() - target) ** 2).mean()
    ent = entropy.mean()
    loss = pg - ec * ent + vl * vc
    stats = torch.stack([loss, vl, pg, ent, ((ratio - 1) - logratio).mean(), (-logratio).mean(),
                         ((ratio - 1.0).abs() > clip).float().mean()])
    return loss, stats


@pytest.mark.parametrize("M,K,indexed", [(1, 7, False), (300, 7, True), (16384, 7, True), (5000, 3, False), (70000, 8, True)])
def test_loss_statistics_and_head_gradients(M, K, indexed):
    from marl_amd.ppo_loss import ppo_loss

    g = torch.Generator(device="cuda").manual_seed(M + K)
    total = 3 * M if indexed else M
    store_actions = torch.randint(0, K, (total,), device="cuda", generator=g).float()
    # old log-probabilities near the new ones, with a tail far outside the clip range
    store_adv = torch.randn(total, device="cuda", generator=g)
    store_adv[::9] = 0.0  # ties between the clipped and the unclipped branch
    store_target = torch.randn(total, device="cuda", generator=g) * 3
    index = torch.randperm(total, device="cuda", generator=g)[:M].contiguous() if indexed else None
    logits = (torch.randn(M, K, device="cuda", generator=g) * 2).requires_grad_()
    values = torch.randn(M, 1, device="cuda", generator=g).requires_grad_()
    sel = index if indexed else torch.arange(M, device="cuda")
    with torch.no_grad():
        lp_now = torch.distributions.Categorical(logits=logits).log_prob(store_actions[sel].long())
    store_old = torch.zeros(total, device="cuda")
    store_old[sel] = lp_now + torch.randn(M, device="cuda", generator=g) * 0.25
    clip, ec, vc = 0.2, 0.01, 0.5

    loss, stats = ppo_loss(logits, values, index, store_actions, store_old, store_adv, store_target, clip, ec, vc)
    (loss * 1.0).backward()
    got_gl, got_gv = logits.grad.clone(), values.grad.clone()
    logits.grad = values.grad = None

    ref_loss, ref_stats = _reference(logits, values, store_actions[sel].long(), store_old[sel], store_adv[sel],
                                     store_target[sel], clip, ec, vc)
    ref_loss.backward()
    # means of M float32 terms: float32 rounding of the reference's own reductions dominates
    torch.testing.assert_close(stats.float(), ref_stats, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(loss, ref_loss, rtol=2e-5, atol=2e-6)
    # per-sample gradients are O(1/M): compare scaled by M
    torch.testing.assert_close(got_gl * M, logits.grad * M, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(got_gv * M, values.grad * M, rtol=1e-5, atol=1e-6)


def test_upstream_gradient_scales_the_head_gradients():
    from marl_amd.ppo_loss import ppo_loss

    torch.manual_seed(0)
    M, K = 64, 7
    logits = torch.randn(M, K, device="cuda", requires_grad=True)
    values = torch.randn(M, 1, device="cuda", requires_grad=True)
    a = torch.randint(0, K, (M,), device="cuda").float()
    z = torch.zeros(M, device="cuda")
    adv = torch.randn(M, device="cuda")
    loss, _ = ppo_loss(logits, values, None, a, z - 1.9, adv, z, 0.2, 0.01, 0.5)
    loss.backward()
    g1 = logits.grad.clone()
    logits.grad = None
    loss, _ = ppo_loss(logits, values, None, a, z - 1.9, adv, z, 0.2, 0.01, 0.5)
    (3.0 * loss).backward()
    torch.testing.assert_close(logits.grad, 3.0 * g1, rtol=1e-6, atol=0)
    with pytest.raises(ValueError):
        ppo_loss(logits, values.squeeze(1), None, a, z, adv, z, 0.2, 0.01, 0.5)

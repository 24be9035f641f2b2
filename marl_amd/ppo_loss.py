"""The clipped-surrogate loss of a PPO minibatch (agents/MAPPO.py:167-200; the caller supplies each
sample's value target, its own for MAPPO, the minibatch mean for CMAPPO)
with its gradients for the two network heads, from libmarl_amd.so `marl_ppo_loss`: two launches
in place of the ~70 small tensor ops autograd runs for the same arithmetic (Categorical
log-probability and entropy, ratio, clipping, the three means, and all their backward ops).
At the reference's batch sizes a minibatch step is launch-bound, not arithmetic-bound."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class _PPOLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, values, index, actions_f, old_logprob, advantages, value_target, clip, entropy_coef,
                value_coef):
        m, k = logits.shape
        logits = logits.contiguous()
        values = values.reshape(m).contiguous()
        for name, t in (("actions", actions_f), ("old_logprob", old_logprob), ("advantages", advantages),
                        ("value_target", value_target)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 1:
                raise ValueError(f"{name} must be a flat contiguous float32 tensor")
        if index is not None and (index.dtype != torch.int64 or index.shape != (m,) or not index.is_contiguous()):
            raise ValueError("index must be a contiguous int64 tensor with one entry per row of logits")
        dev = logits.device
        grad_logits = torch.empty_like(logits)
        grad_values = torch.empty(m, dtype=torch.float32, device=dev)
        stats = torch.empty(7, dtype=torch.float64, device=dev)
        L = _lib.lib()
        work = torch.empty(max(L.marl_ppo_loss_workspace_bytes(m) // 8, 1), dtype=torch.float64, device=dev)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(L.marl_ppo_loss(p(logits), m, k, k, p(values), p(index), p(actions_f), p(old_logprob), p(advantages),
                                       p(value_target), float(clip), float(entropy_coef), float(value_coef),
                                       p(grad_logits), p(grad_values), p(stats), p(work),
                                       ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "marl_ppo_loss")
        ctx.save_for_backward(grad_logits, grad_values)
        ctx.values_shape = None
        ctx.mark_non_differentiable(stats)
        return stats[0].float(), stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        grad_logits, grad_values = ctx.saved_tensors
        return (grad_logits * g_loss, (grad_values * g_loss).unsqueeze(1)) + (None,) * 8


def ppo_loss(logits, values, index, actions_f, old_logprob, advantages, value_target, clip, entropy_coef, value_coef):
    """logits float32 [M, K] and values float32 [M, 1] (both heads' outputs for the minibatch);
    actions_f / old_logprob / advantages / value_target: the rollout store's flat float32 arrays,
    read at `index` (int64 [M]; None: entry n for row n).  Returns (loss, stats) with stats =
    float64 [7]: loss, value loss, policy loss, entropy, approx_kl, old_approx_kl, clip fraction.
    `loss.backward()` feeds the two heads."""
    if values.dim() != 2 or values.shape[1] != 1:
        raise ValueError("values must be [M, 1]")
    return _PPOLoss.apply(logits, values, index, actions_f, old_logprob, advantages, value_target, clip, entropy_coef,
                          value_coef)

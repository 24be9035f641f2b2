"""GAE / value targets on the GPU (libmarl_amd.so `marl_gae`).

Same contract as the reference's ``MAPPO.compute_gae`` (agents/MAPPO.py:82-114) plus the
value-target line (agents/MAPPO.py:144), and ``CMAPPO.compute_gae``
(agents/CentralizedMAPPO.py:48-82) for ``centralised=True``.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _workspace(T, N, device):
    nbytes = int(_lib.lib().marl_gae_workspace_bytes(int(T), int(N)))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise ValueError(f"{name} must be a float32 tensor on the GPU")
    return t.contiguous()


def compute_gae(rewards, dones, values, next_values, gamma=0.99, lam=0.95, centralised=False,
                reward_mean=None, want_returns=True):
    """rewards/dones/values: float32 [T, N] on the GPU; next_values: [1, N] (or [1, 1] when
    centralised).  Returns (advantages, value_targets).  ``reward_mean`` (float32 [T]) lets a
    sharded caller supply the all-rank mean of the rewards for the centralised variant."""
    rewards, dones, values = _f32c(rewards, "rewards"), _f32c(dones, "dones"), _f32c(values, "values")
    T, N = rewards.shape
    if dones.shape != (T, N) or values.shape != (T, N):
        raise ValueError("rewards, dones and values must share the shape [T, N]")
    next_values = _f32c(next_values, "next_values").reshape(-1)
    if next_values.numel() != (1 if centralised else N):
        raise ValueError(f"next_values has {next_values.numel()} entries, expected {1 if centralised else N}")
    adv = torch.empty_like(rewards)
    ret = torch.empty_like(rewards) if want_returns else None
    ready = 0
    workspace = None
    if centralised:
        workspace = _workspace(T, N, rewards.device)
        if reward_mean is not None:
            reward_mean = _f32c(reward_mean, "reward_mean")
            if reward_mean.numel() != T:
                raise ValueError("reward_mean must have T entries")
            workspace[: T * 4].view(torch.float32).copy_(reward_mean.reshape(-1))
            ready = 1
    stream = ctypes.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)
    with torch.cuda.device(rewards.device):
        _lib.check(_lib.lib().marl_gae(_ptr(rewards), _ptr(dones), _ptr(values), _ptr(next_values), T, N,
                                       float(gamma), float(lam), _ptr(adv), _ptr(ret), 1 if centralised else 0,
                                       _ptr(workspace), ready, stream), "marl_gae")
    return adv, ret


def row_sums(x):
    """float64 [T] sums over the rows of a float32 [T, N] GPU tensor (fixed order, reproducible)."""
    x = _f32c(x, "x")
    T, N = x.shape
    out = torch.empty(T, dtype=torch.float64, device=x.device)
    workspace = _workspace(T, N, x.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().marl_row_sums(_ptr(x), T, N, _ptr(out), _ptr(workspace), stream), "marl_row_sums")
    return out

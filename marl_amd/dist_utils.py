"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" is RCCL on
ROCm; "gloo" for CPU tests).  Envs are sharded by index and never exchanged; the only
collectives are one all-reduce of the flattened gradient per minibatch (placed between the
reference's ``loss.backward()`` and ``clip_grad_norm_``, agents/MAPPO.py:207-208), small
episode-statistic reductions, and for centralised MAPPO the per-step reward mean."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* if the job was
    launched by torch.distributed.run; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # MARL_AMD_DIST_BACKEND=gloo lets several ranks share one GPU (tests); RCCL needs one GPU per rank
            backend = os.environ.get("MARL_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kwargs)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class FlatGradAllReducer:
    """Averages gradients across ranks with ONE collective per call: the gradients that exist
    are packed into a persistent flat buffer (~0.95 MB fp32 for the 202-input policy),
    all-reduced, and unpacked in place.  At this size the collective is latency-bound, so one
    call beats one per parameter.  Parameters without a gradient (the centralised critic is
    never trained, model.py:56-60, and is as wide as D * A * E) take no part: every rank runs
    the same graph, so the set is the same everywhere."""

    def __init__(self, parameters):
        self.params = [p for p in parameters if p.requires_grad]
        self.active = None
        self.flat = None

    @torch.no_grad()
    def __call__(self):
        w = world_size()
        if w == 1:
            return
        active = [k for k, p in enumerate(self.params) if p.grad is not None]
        if not active:
            return
        if active != self.active:
            ref = self.params[active[0]]
            n = sum(self.params[k].numel() for k in active)
            self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
            self.active = active
        off = 0
        for k in active:
            g = self.params[k].grad
            self.flat[off:off + g.numel()].copy_(g.reshape(-1))
            off += g.numel()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(w)
        off = 0
        for k in active:
            g = self.params[k].grad
            g.copy_(self.flat[off:off + g.numel()].reshape(g.shape))
            off += g.numel()


def all_reduce_sum_(t):
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_parameters(module, src=0):
    if world_size() > 1:
        for p in module.parameters():
            dist.broadcast(p.data, src)
        for b in module.buffers():
            dist.broadcast(b.data, src)

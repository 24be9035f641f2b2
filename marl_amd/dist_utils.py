"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" is RCCL on
ROCm; "gloo" for CPU tests).  Envs are sharded by index and never exchanged; the only
collectives are one all-reduce of the flattened gradient per minibatch (placed between the
reference's ``loss.backward()`` and ``clip_grad_norm_``, agents/MAPPO.py:207-208), small
episode-statistic reductions, and for centralised MAPPO the per-step reward mean."""
from __future__ import annotations

import atexit
import os

import torch
import torch.distributed as dist


def _forced():
    """MARL_AMD_FORCE_COLLECTIVES=1: issue every collective even in a one-rank group, so that a single
    GPU exercises the RCCL code path end to end (tests)."""
    return os.environ.get("MARL_AMD_FORCE_COLLECTIVES") == "1"


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* if the job was
    launched by torch.distributed.run; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or (_forced() and "RANK" in os.environ)) and not dist.is_initialized():
        if backend is None:
            # MARL_AMD_DIST_BACKEND=gloo lets several ranks share one GPU (tests); RCCL needs one GPU per rank
            backend = os.environ.get("MARL_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kwargs)
        atexit.register(_shutdown)
    return rank, local_rank, world


def _shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def collectives_active():
    """True when collectives have to be issued: more than one rank, or a one-rank group with
    MARL_AMD_FORCE_COLLECTIVES=1."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or _forced()


def backend_name():
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


_trace_count = 0


def _reduce(t):
    """SUM over ranks, in place.  RCCL reduces device memory on its own stream, ordered behind the
    caller's stream by an event; nothing waits on the host.  gloo cannot reduce device memory: given a
    device tensor it stages it through the host itself, from a worker thread that waits on a side
    stream ordered behind the caller's stream, while the calling thread blocks until that worker is
    done.  On ROCm 7.2 that wait never returned once the caller's stream had replayed hipGraphs
    (two ranks sharing a GPU, second graph-replayed rollout: both ranks stuck in the same
    all_reduce, worker thread in its stream wait; profiles/diag/two_rank_train.sh), whereas a wait
    on the caller's own stream from the calling thread always returned.  So for gloo the staging
    is done here, on the caller's stream: copy to the host (which waits for that stream), reduce
    host memory, copy back."""
    if t.is_cuda and backend_name() == "gloo" and os.environ.get("MARL_AMD_GLOO_DEVICE_TENSORS") != "1":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
        return
    dist.all_reduce(t, op=dist.ReduceOp.SUM)


def _all_reduce(t):
    """_reduce; MARL_AMD_DIST_TRACE=1 prints entry and exit per rank (diagnosis of stalls)."""
    global _trace_count
    if os.environ.get("MARL_AMD_DIST_TRACE") == "1":
        import sys

        _trace_count += 1
        print(f"[rank {get_rank()}] all_reduce #{_trace_count} {tuple(t.shape)} {t.dtype} on {t.device}: enter",
              file=sys.stderr, flush=True)
        _reduce(t)
        print(f"[rank {get_rank()}] all_reduce #{_trace_count}: returned", file=sys.stderr, flush=True)
        return
    _reduce(t)


class FlatGradAllReducer:
    """Averages gradients across ranks with ONE collective per call: the gradients that exist
    are packed into a persistent flat buffer (~0.95 MB fp32 for the 202-input policy),
    all-reduced, and unpacked in place.  At this size the collective is latency-bound, so one
    call beats one per parameter.  Packing and unpacking are one multi-tensor launch each
    (``torch._foreach_copy_`` between the gradients and same-shaped views of the flat buffer): two
    launches around the collective whatever the number of parameters.  (Making the ``.grad``s
    themselves views of the buffer would save those two but cost more: autograd then accumulates
    into the existing gradient with one add per parameter and the buffer needs a zero-fill per
    step.)  Parameters without a gradient (the centralised critic is never trained,
    model.py:56-60, and is as wide as D * A * E) take no part: every rank runs the same graph,
    so the set is the same everywhere."""

    def __init__(self, parameters):
        self.params = [p for p in parameters if p.requires_grad]
        self.active = None
        self.flat = None
        self.views = None
        self.calls = 0  # collectives issued (tests)

    @torch.no_grad()
    def __call__(self):
        if not collectives_active():
            return
        w = world_size()
        active = [k for k, p in enumerate(self.params) if p.grad is not None]
        if not active:
            return
        if active != self.active:
            ref = self.params[active[0]]
            n = sum(self.params[k].numel() for k in active)
            self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
            self.views, off = [], 0
            for k in active:
                p = self.params[k]
                self.views.append(self.flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self.active = active
        grads = [self.params[k].grad for k in active]
        torch._foreach_copy_(self.views, grads)
        _all_reduce(self.flat)
        self.calls += 1
        if w > 1:
            self.flat.div_(w)
        torch._foreach_copy_(grads, self.views)


def all_reduce_sum_(t):
    if collectives_active():
        _all_reduce(t)
    return t


def broadcast_parameters(module, src=0):
    if collectives_active():
        for p in module.parameters():
            dist.broadcast(p.data, src)
        for b in module.buffers():
            dist.broadcast(b.data, src)

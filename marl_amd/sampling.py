"""Categorical sampling of the rollout on the GPU (libmarl_amd.so `marl_sample_categorical`):
the sampling half of ``MAPPO.act`` (agents/MAPPO.py:58-77, model.py:52-56) as one launch
instead of the dozen small tensor ops behind ``Categorical(logits=...).sample()``,
``.log_prob()`` and ``.entropy()``."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class CategoricalSampler:
    """Draws from Categorical(logits=row) for every row.  The uniform behind row r of call c
    comes from a counter RNG keyed by (seed, c, row_index_base + r); the call counter lives on
    the device and is advanced by a captured op, so a launch replayed from a hipGraph keeps
    drawing fresh numbers."""

    def __init__(self, seed: int, row_index_base: int = 0, device="cuda"):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.row_index_base = int(row_index_base)
        self.device = torch.device(device)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._slots = None

    def use_slots(self, calls_per_round: int) -> None:
        """For a loop that makes the same `calls_per_round` calls round after round (a rollout):
        call k of a round reads its call number from its own device word, and `advance_round()`
        moves all of them on at once, so the per-call increment launch goes away.  The numbers
        are those the single counter would have taken (round * calls_per_round + k)."""
        if self._slots is None or self._slots.numel() != calls_per_round:
            base = int(self.counter.item())
            self._slots = base + torch.arange(calls_per_round, dtype=torch.int64, device=self.device)

    def advance_round(self) -> None:
        self._slots.add_(self._slots.numel())
        self.counter.add_(self._slots.numel())

    def sample(self, logits: torch.Tensor, want_entropy: bool = True, actions_out=None, logprob_out=None, slot=None,
               num_actions=None, next_column_out=None, row_offset=0):
        """`actions_out` (int64 [rows]) / `logprob_out` (float32 [rows]): contiguous tensors to write
        into, e.g. a row of the rollout store, instead of fresh ones.  `slot`: index of this call
        within its round (see `use_slots`); without it the single counter is used and advanced.
        `num_actions`: the first so many columns of `logits` are the logits (default: all);
        `next_column_out` (float32 [rows]) then receives the column after them, e.g. the critic's
        value when both heads come out of one GEMM.  `row_offset`: `logits` holds rows
        [row_offset, row_offset + n) of the batch (a shard of it): they draw the numbers they would
        draw in a call over the whole batch."""
        if logits.dtype != torch.float32 or not logits.is_cuda or logits.dim() != 2:
            raise ValueError("logits must be a float32 [rows, actions] tensor on the GPU")
        logits = logits.contiguous()
        n, ld = logits.shape
        k = ld if num_actions is None else int(num_actions)
        if not 0 < k <= ld or (next_column_out is not None and k >= ld):
            raise ValueError(f"num_actions={k} does not fit rows of {ld} columns (next_column_out needs one more)")
        for t, dt, name in ((actions_out, torch.int64, "actions_out"), (logprob_out, torch.float32, "logprob_out"),
                            (next_column_out, torch.float32, "next_column_out")):
            if t is not None and (t.dtype != dt or t.shape != (n,) or not t.is_contiguous() or t.device != logits.device):
                raise ValueError(f"{name} must be a contiguous {dt} tensor of {n} entries on the logits' device")
        actions = actions_out if actions_out is not None else torch.empty(n, dtype=torch.int64, device=logits.device)
        logprob = logprob_out if logprob_out is not None else torch.empty(n, dtype=torch.float32, device=logits.device)
        entropy = torch.empty(n, dtype=torch.float32, device=logits.device) if want_entropy else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        counter = self.counter if slot is None else self._slots[slot]
        _lib.check(_lib.lib().marl_sample_categorical(p(logits), n, k, ld, self.seed, p(counter),
                                                      self.row_index_base + int(row_offset),
                                                      p(actions), p(logprob), p(entropy), p(next_column_out), stream),
                   "marl_sample_categorical")
        if slot is None:
            self.counter.add_(1)
        return actions, logprob, entropy

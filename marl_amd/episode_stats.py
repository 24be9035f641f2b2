"""Reward-event bookkeeping of the rollout driver on the GPU (libmarl_amd.so
`marl_reward_events`), counterpart of agent_environment.py:66-76.

The reference classifies every step from the float32 rewards of rows 0..1 (a delivery when any
is >= 1, a plating when any == 0.3, an onion-in-pot when any == 0.1) and adds their mean to the
episode return.  `EpisodeStats` keeps exactly that for rows 0..1 and the same per env for every
env, as integer totals on the device; the host reads them once per episode.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class EpisodeStats:
    MAX_STEPS_PER_CALL = 4096  # MARL_REWARD_EVENTS_MANY_MAX_STEPS of include/marl_amd.h

    def __init__(self, num_envs: int, num_agents: int, device):
        self.E, self.A = int(num_envs), int(num_agents)
        self.device = torch.device(device)
        # int64 storage of the uint64 totals (the values stay far below 2^63)
        self.totals = torch.zeros(4, dtype=torch.int64, device=self.device)
        self.first_rows = torch.zeros(4, dtype=torch.float64, device=self.device)
        nbytes = int(_lib.lib().marl_reward_events_workspace_bytes(self.E, self.A))
        self._workspace = torch.empty(max(nbytes, 8) // 8, dtype=torch.int64, device=self.device)
        self.steps = 0
        self._many_workspace = None

    def add(self, rewards: torch.Tensor) -> None:
        """Adds one step's rewards (float32 [E*A] on the GPU, env-major)."""
        if rewards.dtype != torch.float32 or not rewards.is_cuda or not rewards.is_contiguous():
            raise ValueError("rewards must be a contiguous float32 tensor on the GPU")
        if rewards.numel() != self.E * self.A:
            raise ValueError(f"rewards has {rewards.numel()} entries, expected {self.E * self.A}")
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().marl_reward_events(
                ctypes.c_void_p(rewards.data_ptr()), self.E, self.A, ctypes.c_void_p(self.totals.data_ptr()),
                ctypes.c_void_p(self.first_rows.data_ptr()), ctypes.c_void_p(self._workspace.data_ptr()), stream),
                "marl_reward_events")
        self.steps += 1

    def add_many(self, rewards: torch.Tensor) -> None:
        """Adds several steps' rewards at once (float32 [K, E*A], contiguous): the same totals as K
        `add` calls in order, from two launches instead of up to 2 K."""
        if rewards.dtype != torch.float32 or not rewards.is_cuda or not rewards.is_contiguous() or rewards.dim() != 2:
            raise ValueError("rewards must be a contiguous float32 [steps, rows] tensor on the GPU")
        K = int(rewards.shape[0])
        if rewards.shape[1] != self.E * self.A:
            raise ValueError(f"rewards has {rewards.shape[1]} entries per step, expected {self.E * self.A}")
        if K == 0:
            return
        if K > self.MAX_STEPS_PER_CALL:  # the library takes a rollout's worth of steps per call
            for first in range(0, K, self.MAX_STEPS_PER_CALL):
                self.add_many(rewards[first:first + self.MAX_STEPS_PER_CALL])
            return
        L = _lib.lib()
        need = max(int(L.marl_reward_events_many_workspace_bytes(K, self.E, self.A)), 8) // 8
        if self._many_workspace is None or self._many_workspace.numel() < need:
            self._many_workspace = torch.empty(need, dtype=torch.int64, device=self.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(L.marl_reward_events_many(
                ctypes.c_void_p(rewards.data_ptr()), K, self.E, self.A, ctypes.c_void_p(self.totals.data_ptr()),
                ctypes.c_void_p(self.first_rows.data_ptr()), ctypes.c_void_p(self._many_workspace.data_ptr()), stream),
                "marl_reward_events_many")
        self.steps += K

    def raw(self):
        """(totals int64[4], first_rows float64[4]) as device tensors, without a host sync."""
        return self.totals, self.first_rows

    def read(self) -> dict:
        """One host sync.  `first_rows_*`: the reference's statistics (rows 0..1); `all_*`: per-env
        averages over every env of this process."""
        t = self.totals.tolist()
        f = self.first_rows.tolist()
        return {
            "first_rows_return": f[0], "first_rows_deliveries": f[1], "first_rows_platings": f[2],
            "first_rows_pottings": f[3],
            "all_return": t[0] / 10.0 / (self.E * self.A), "all_deliveries": t[1] / self.E,
            "all_platings": t[2] / self.E, "all_pottings": t[3] / self.E,
        }

    def reset(self) -> None:
        self.totals.zero_()
        self.first_rows.zero_()
        self.steps = 0

"""Fixed-length on-device rollout store, interface of the reference's ``Buffer``
(buffer.py:3-35): six float32 tensors shaped (T, N[, D]) with N = num_env * num_agents,
``add`` writes row ``ptr``, ``reset`` rewinds.  Actions and done flags are stored as
float32 exactly like the reference does (buffer.py:12,14)."""
from __future__ import annotations

import torch


class Buffer:
    def __init__(self, state_dim, num_agents, num_env, max_size=128, centralized=False, device=None):
        self.max_size = int(max_size)
        self.ptr = 0
        self.size = 0
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        rows = int(num_env) * int(num_agents)
        mk = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)  # noqa: E731
        self.obs_buff = mk(self.max_size, rows, int(state_dim))
        self.actions_buff = mk(self.max_size, rows)
        self.rewards_buff = mk(self.max_size, rows)
        self.dones_buff = mk(self.max_size, rows)
        self.logprobs_buff = mk(self.max_size, rows)
        # buffer.py:17-18 keeps a (T, 1) variant that no caller of the reference enables
        self.values_buff = mk(self.max_size, 1) if centralized else mk(self.max_size, rows)

    def obs_slot(self, index=None):
        """The (N, D) view the next ``add`` would fill: lets the env kernel write straight
        into the rollout store instead of producing a tensor that is copied afterwards."""
        return self.obs_buff[self.ptr if index is None else index]

    def add(self, obs, actions, rewards, dones, logprobs, values):
        p = self.ptr
        if obs.data_ptr() != self.obs_buff[p].data_ptr():
            self.obs_buff[p] = obs
        self.actions_buff[p] = actions
        self.rewards_buff[p] = rewards
        self.dones_buff[p] = dones
        self.logprobs_buff[p] = logprobs
        self.values_buff[p] = values
        self.ptr = (p + 1) % self.max_size
        self.size = min(self.size + 1, self.max_size)

    def reset(self):
        self.ptr = 0
        self.size = 0

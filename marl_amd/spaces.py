"""Minimal stand-ins for the gymnasium spaces the reference reads
(`single_observation_space.shape`, `single_action_space.n/.shape`: main.py:92-93,195-196,
model.py:20,25, agents/MAPPO.py:147-149)."""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


class Discrete:
    def __init__(self, n: int):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def __repr__(self):
        return f"Discrete({self.n})"


class Dict(dict):
    """Dict observation space: a mapping from key to space."""

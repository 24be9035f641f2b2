"""One batch of Overcooked envs as K shards on K HIP streams of one GPU.

A single fused step launch over E envs starts with every wave simulating (nothing is written for the
first few microseconds) and ends with a drain; at 65,536 envs the launch is one round of waves, so
neither hides behind the row stream.  Two independent half-batch launches on two streams do hide
each other's start and end: one shard simulates while the other streams its rows.  The shards are
ordinary env handles whose ``env_index_base`` continues the batch's numbering, so shard k
reproduces exactly envs [k*E/K, (k+1)*E/K) of the single-handle batch (same RNG keys), and its
rows are the contiguous slice ``rows(k)`` of every [N, ...] tensor of the batch.

Two ways to drive it:
  * ``reset`` / ``step`` / ``step_into`` keep the vec-env surface of ``BatchedOvercookedVecEnv``
    (what ``agent_environment.py:34,50,88`` calls): the shards are forked from and joined back to
    the caller's stream inside every call;
  * ``run_pipelined(steps, body)`` forks once, issues ``body(k, shard, rows, step)`` for every step
    on shard k's stream, and joins once at the end: no join between steps, so the shards drift out
    of phase.  A consumer that needs a step's rows has to sit on the shard's stream too (a policy
    forward per shard: ``rollout.agent_environment_loop`` does that).  Captured into a hipGraph
    this becomes one branch per shard.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .env import OBS_KEY, BatchedOvercookedVecEnv
from .spaces import Dict


class ShardedVecEnv:
    def __init__(self, layout_rows, num_agents, feature="global_obs", num_envs=2, shards=2, max_steps=1000, seed=0,
                 env_index_base=0, device=None, dist_metric=0, auto_reset=True):
        K = int(shards)
        if K < 1 or int(num_envs) % K:
            raise ValueError(f"num_envs={num_envs} is not a multiple of shards={shards}")
        if not torch.cuda.is_available():
            raise _lib.MarlAmdError("ShardedVecEnv needs a HIP device (there is no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.E, self.A, self.K = int(num_envs), int(num_agents), K
        per = self.E // K
        self.shards = [BatchedOvercookedVecEnv(layout_rows, num_agents, feature, per, max_steps=max_steps, seed=seed,
                                               env_index_base=int(env_index_base) + k * per, device=self.device,
                                               dist_metric=dist_metric, auto_reset=auto_reset) for k in range(K)]
        first = self.shards[0]
        self.feature, self.max_steps, self.layout_rows = first.feature, first.max_steps, first.layout_rows
        self.seed, self.env_index_base = int(seed), int(env_index_base)
        self.D, self.NP, self.state_words, self.H, self.W = first.D, first.NP, first.state_words, first.H, first.W
        self.N = self.E * self.A
        self.rows_per_shard = per * self.A
        # the env kernel writes a shard's rows at a 16-byte aligned address
        if K > 1 and (self.rows_per_shard * self.D) % 4:
            raise ValueError(f"{per} envs per shard x {self.A} agents x D={self.D}: a shard's rows must be a multiple "
                             "of 16 bytes for the shards of one tensor to start aligned")
        self.num_envs = self.N  # rows, as the reference's vec env counts (main.py:186)
        self.single_observation_space = first.single_observation_space
        self.single_action_space = first.single_action_space
        self.observation_space = Dict({OBS_KEY: self.single_observation_space})
        self.action_space = self.single_action_space
        self.numpy_io = False
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(K)]

    # ------------------------------------------------------------------
    @property
    def num_shards(self):
        return self.K

    def rows(self, k):
        """the slice of every [N, ...] tensor of the batch that belongs to shard k"""
        return slice(k * self.rows_per_shard, (k + 1) * self.rows_per_shard)

    def fork(self):
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(main)

    def join(self):
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            main.wait_stream(s)

    def run_pipelined(self, steps, body):
        """``body(k, shard, rows, step)`` for every step in ``steps`` and every shard k, issued on shard
        k's stream; one fork from the caller's stream before the first step and one join after the
        last, none in between.  Works eagerly and under hipGraph capture (a branch per shard)."""
        self.fork()
        for step in steps:
            for k, sh in enumerate(self.shards):
                with torch.cuda.stream(self.streams[k]):
                    body(k, sh, self.rows(k), step)
        self.join()

    # ------------------------------------------------------------------
    def _check_rows(self, t, name, trailing=()):
        if t is None or tuple(t.shape) != (self.N,) + tuple(trailing) or not t.is_contiguous() or t.device != self.device:
            raise ValueError(f"{name} must be a contiguous tensor of shape {(self.N,) + tuple(trailing)} on {self.device}")

    def _new_obs(self, out):
        if out is None:
            return torch.empty((self.N, self.D), dtype=torch.float32, device=self.device)
        self._check_rows(out, "out", (self.D,))
        if out.dtype != torch.float32 or out.data_ptr() % 16:
            raise ValueError("out must be float32 and 16-byte aligned")
        return out

    def reset(self, seed=None, options=None, out=None):
        """agent_environment.py:34,88: every env restarts (all shards, joined)."""
        obs = self._new_obs(out)
        if seed is not None:
            self.seed = int(seed)
        self.fork()
        for k, sh in enumerate(self.shards):
            with torch.cuda.stream(self.streams[k]):
                sh.reset(seed=seed, out=obs[self.rows(k)])
        self.join()
        return {OBS_KEY: obs}, {}

    def step_into(self, actions, obs, rewards, terminated, truncated):
        """One step of every shard, forked from and joined to the caller's stream: the call has the
        stream semantics of ``BatchedOvercookedVecEnv.step_into`` (and its results)."""
        self.fork()
        for k, sh in enumerate(self.shards):
            r = self.rows(k)
            with torch.cuda.stream(self.streams[k]):
                sh.step_into(actions[r], obs[r], rewards[r], terminated[r], truncated[r])
        self.join()

    def step(self, actions, out=None):
        """agent_environment.py:50."""
        act = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions))
        act = act.to(self.device)
        if act.dtype not in (torch.int64, torch.int32, torch.uint8):
            act = act.to(torch.int64)
        act = act.contiguous()
        if act.numel() != self.N:
            raise ValueError(f"actions must have {self.N} entries, got {tuple(act.shape)}")
        obs = self._new_obs(out)
        rew = torch.empty(self.N, dtype=torch.float32, device=self.device)
        term = torch.empty(self.N, dtype=torch.bool, device=self.device)
        trunc = torch.empty(self.N, dtype=torch.bool, device=self.device)
        self.step_into(act.view(-1), obs, rew, term, trunc)
        return {OBS_KEY: obs}, rew, term, trunc, {}

    def featurize(self, out=None):
        obs = self._new_obs(out)
        self.fork()
        for k, sh in enumerate(self.shards):
            with torch.cuda.stream(self.streams[k]):
                sh.featurize(out=obs[self.rows(k)])
        self.join()
        return obs

    def export_state(self):
        torch.cuda.current_stream(self.device).synchronize()
        snaps = [sh.export_state() for sh in self.shards]
        return {key: np.concatenate([s[key] for s in snaps], axis=0) for key in snaps[0]}

    def close(self):
        for sh in self.shards:
            sh.close()

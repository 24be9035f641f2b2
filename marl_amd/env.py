"""Batched Overcooked vector env on MI355X behind the reference's vec-env surface.

Mirrors what the reference's drivers touch:
  * ``make_env(num_agents, layout, feature, render_mode)``           main.py:29-64
  * ``make_vector_env(num_envs, overcooked_env)``                     main.py:66-96
  * ``vec.num_envs == E * A``, ``vec.single_observation_space.shape == (D,)``,
    ``vec.single_action_space.n == 7``                                main.py:92-93,186,195-196
  * ``vec.reset() -> (obs_dict, info)``, ``vec.step(actions) ->
    (obs_dict, rewards, terminated, truncated, info)`` with rows env-major and observations
    under ``'n_agent_overcooked_features'``                           agent_environment.py:34-36,50-53,88,106-107

All simulation and featurisation happens in libmarl_amd.so (HIP, gfx950); this module only
owns device buffers (torch tensors) and argument checking.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .layouts import get_layout
from .spaces import Box, Dict, Discrete

OBS_KEY = "n_agent_overcooked_features"  # overcooked_features.py:103

FEATURE_IDS = {
    "global_obs": 0,                          # overcooked_config.py:14-16
    "local_obs": 1,                           # overcooked_config.py:18-20
    "Minimal_spatial": 2,                     # overcooked_config.py:26-28
    "Minimal_spatial_other_agent_aware": 3,   # overcooked_config.py:22-24
    "only_direction": 4,                      # README.md:65 (AgentDir only, README.md:112)
}
NUM_ACTIONS = 7  # cardinal_actions, overcooked_config.py:79


def block_dim(feature: str, num_agents: int) -> int:
    fid = FEATURE_IDS[feature]
    if fid in (0, 1):
        return 99 + 2 * (num_agents - 1)
    if fid == 2:
        return 33
    if fid == 3:
        return 33 + 2 * (num_agents - 1)
    return 4


def obs_dim(feature: str, num_agents: int) -> int:
    b = block_dim(feature, num_agents)
    return num_agents * b if feature == "global_obs" else b


class OvercookedEnvSpec:
    """What ``make_env`` returns: the single-env description the reference reads
    (``observation_spaces[i][OBS_KEY].shape``, ``action_spaces[i].n``, ``config["num_agents"]``:
    main.py:206-207,295)."""

    def __init__(self, num_agents: int, layout: str, feature: str, max_steps: int = 1000, render_mode=None):
        if feature not in FEATURE_IDS:
            raise KeyError(f"unknown feature {feature!r}; known: {sorted(FEATURE_IDS)}")
        self.layout_rows = get_layout(layout)
        # overcooked_config.py:68-110 with main.py:47-50 applied
        self.config = {
            "name": "NAgentOvercooked-V0",
            "num_agents": int(num_agents),
            "action_set": "cardinal_actions",
            "features": feature,
            "rewards": ["onion_in_pot_reward", "soup_in_dish_reward", "delivery_reward"],
            "scope": "overcooked",
            "grid": {"layout": layout},
            "max_steps": int(max_steps),
        }
        self.render_mode = render_mode
        self.agent_ids = list(range(num_agents))
        self.possible_agents = list(self.agent_ids)
        d = obs_dim(feature, num_agents)
        self.observation_spaces = {
            i: Dict({OBS_KEY: Box(-np.inf, np.inf, (d,), np.float32)}) for i in self.agent_ids
        }
        self.action_spaces = {i: Discrete(NUM_ACTIONS) for i in self.agent_ids}
        self._single = None
        self._seed = 0

    # -- single-env stepping with per-agent dicts, the shape tests/test_load.py:74-79,104-128 uses --
    def _env1(self):
        if self._single is None:
            self._single = BatchedOvercookedVecEnv(self.layout_rows, self.config["num_agents"],
                                                   self.config["features"], 1, max_steps=self.config["max_steps"],
                                                   seed=self._seed, numpy_io=True, auto_reset=False)
        return self._single

    def reset(self, seed=None, options=None):
        if seed is not None and (self._single is None or seed != self._seed):
            self._seed, self._single = int(seed), None
        obs, _ = self._env1().reset()
        return {i: {OBS_KEY: obs[OBS_KEY][i]} for i in self.agent_ids}, {i: {} for i in self.agent_ids}

    def step(self, actions):
        """actions: {agent_id: int}.  Returns per-agent dicts (obs, rewards, terminated, truncated, infos);
        no restart happens inside the call (a bare env, unlike the vector wrapper)."""
        act = np.array([int(actions[i]) for i in self.agent_ids], np.int64)
        obs, rew, term, trunc, _ = self._env1().step(act)
        ids = self.agent_ids
        return ({i: {OBS_KEY: obs[OBS_KEY][i]} for i in ids}, {i: float(rew[i]) for i in ids},
                {i: bool(term[i]) for i in ids}, {i: bool(trunc[i]) for i in ids}, {i: {} for i in ids})


def make_env(num_agents=4, layout="large_overcooked_layout", feature="global_obs", render_mode=None, max_steps=1000):
    """main.py:29-64."""
    return OvercookedEnvSpec(num_agents, layout, feature, max_steps=max_steps, render_mode=render_mode)


def make_vector_env(num_envs, overcooked_env: OvercookedEnvSpec, shards=1, **kwargs):
    """main.py:66-96: E copies of the env, agents flattened to rows, one reset issued.
    `shards` > 1: the batch as that many env handles on streams of their own
    (`sharded_env.ShardedVecEnv`; same envs, same results, rollouts can be pipelined)."""
    if int(shards) > 1:
        from .sharded_env import ShardedVecEnv

        vec = ShardedVecEnv(overcooked_env.layout_rows, overcooked_env.config["num_agents"],
                            overcooked_env.config["features"], num_envs, shards=int(shards),
                            max_steps=overcooked_env.config["max_steps"], **kwargs)
        vec.reset()
        return vec
    vec = BatchedOvercookedVecEnv(
        overcooked_env.layout_rows,
        overcooked_env.config["num_agents"],
        overcooked_env.config["features"],
        num_envs,
        max_steps=overcooked_env.config["max_steps"],
        **kwargs,
    )
    vec.reset()
    return vec


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class BatchedOvercookedVecEnv:
    """E Overcooked envs with A agents each, resident on one GPU.

    ``numpy_io=True`` makes reset/step accept and return numpy arrays exactly like the
    reference's multiprocess vector env (so ``agent_environment.py``-shaped loops run
    unchanged); the default hands out device tensors and never leaves the GPU.
    """

    def __init__(self, layout_rows, num_agents, feature="global_obs", num_envs=1, max_steps=1000, seed=0,
                 env_index_base=0, device=None, numpy_io=False, dist_metric=0, auto_reset=True,
                 reuse_host_buffers=False):
        if isinstance(layout_rows, str):
            layout_rows = get_layout(layout_rows)
        if feature not in FEATURE_IDS:
            raise KeyError(f"unknown feature {feature!r}; known: {sorted(FEATURE_IDS)}")
        if not torch.cuda.is_available():
            raise _lib.MarlAmdError("BatchedOvercookedVecEnv needs a HIP device (there is no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._L = _lib.lib()
        self.E, self.A = int(num_envs), int(num_agents)
        self.feature = feature
        self.max_steps = int(max_steps)
        self.numpy_io = bool(numpy_io)
        self.reuse_host_buffers = bool(reuse_host_buffers)
        self.layout_rows = list(layout_rows)
        self.H, self.W = len(layout_rows), len(layout_rows[0])
        self.seed, self.env_index_base, self.dist_metric = int(seed), int(env_index_base), int(dist_metric)
        self.auto_reset = bool(auto_reset)
        self._h = None
        self._create()
        self.D = self._L.marl_env_obs_dim(self._h)
        self.NP = self._L.marl_env_num_pots(self._h)
        self.state_words = self._L.marl_env_state_words(self._h)
        self.N = self.E * self.A
        assert self.D == obs_dim(feature, self.A)
        # the reference's vec env reports rows, not games, as num_envs (main.py:186)
        self.num_envs = self.N
        self.single_observation_space = Box(-np.inf, np.inf, (self.D,), np.float32)
        self.single_action_space = Discrete(NUM_ACTIONS)
        self.observation_space = Dict({OBS_KEY: self.single_observation_space})
        self.action_space = self.single_action_space
        self.state = torch.zeros((self.state_words, self.E), dtype=torch.int32, device=self.device)
        self._closed = False
        self._host_bufs, self._host_flip, self._obs_flip = {}, 0, 0

    def _create(self):
        rows = (ctypes.c_char_p * len(self.layout_rows))(*[r.encode() for r in self.layout_rows])
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.marl_env_create(ctypes.byref(handle), rows, len(self.layout_rows), self.A,
                                               FEATURE_IDS[self.feature], self.max_steps, self.E, self.seed,
                                               self.env_index_base, self.dist_metric), "marl_env_create")
        if self._h:
            self._L.marl_env_destroy(self._h)
        self._h = handle
        if not self.auto_reset:
            _lib.check(self._L.marl_env_set_auto_reset(self._h, 0), "marl_env_set_auto_reset")

    # ------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _to_host(self, name, t):
        """Device tensor -> numpy array through page-locked staging memory (a pageable
        `.cpu()` moves the 106 MB of a 65,536-env observation at ~4 GB/s).  By default the
        caller gets a fresh array per call, as from the reference's vec env.  With
        ``reuse_host_buffers=True`` it gets the staging memory itself: two buffers alternate per
        output, so an array stays valid until the call after the next one -- enough for a loop
        that copies what it receives at once, as the reference's does
        (agent_environment.py:36,106) -- and the copy out of the staging memory is saved."""
        key = (name, self._host_flip if name != "obs" else self._obs_flip)
        if name == "obs":
            self._obs_flip ^= 1
        buf = self._host_bufs.get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._host_bufs[key] = buf
        buf.copy_(t, non_blocking=True)  # the caller synchronises once after queueing all its copies
        return buf.numpy()

    def _new_obs(self, out):
        if out is None:
            return torch.empty((self.N, self.D), dtype=torch.float32, device=self.device)
        if out.shape != (self.N, self.D) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError(f"out must be a contiguous float32 ({self.N}, {self.D}) tensor on {self.device}")
        if out.data_ptr() % 16:
            raise ValueError("out must be 16-byte aligned")
        return out

    def _u8(self, x, name):
        if x is None:
            return None
        if not torch.is_tensor(x):
            # host arrays are checked here; a device tensor is taken as it is (no read-back on this
            # path) and the kernel keeps whatever it names inside its tables
            a = np.asarray(x)
            if a.size != self.N:
                raise ValueError(f"{name} must have {self.E}x{self.A} entries")
            a = a.reshape(self.E, self.A)
            if name == "dir_override":
                if ((a < 0) | (a > 3)).any():
                    raise ValueError("dir_override holds a direction outside 0..3")
            elif name == "spawn_override":
                grid = np.array([list(r) for r in self.layout_rows])
                floor = np.isin(grid, [" ", "+"]).reshape(-1)
                if ((a < 0) | (a >= floor.size)).any() or not floor[a].all():
                    raise ValueError("spawn_override names a cell that is not floor")
                if self.A > 1 and (np.sort(a, axis=1)[:, 1:] == np.sort(a, axis=1)[:, :-1]).any():
                    raise ValueError("spawn_override puts two agents on one cell")
            x = a
        t = torch.as_tensor(x).to(self.device, torch.uint8).contiguous()
        if t.numel() != self.N:
            raise ValueError(f"{name} must have {self.E}x{self.A} entries")
        return t

    def _wrap_obs(self, obs):
        return {OBS_KEY: self._to_host("obs", obs) if self.numpy_io else obs}

    # ------------------------------------------------------------------
    def reset(self, seed=None, options=None, out=None, spawn_override=None, dir_override=None):
        """agent_environment.py:34,88: every env restarts.  The reference never passes a seed;
        when one is given the env RNG is re-keyed with it and its draw counters start over."""
        if seed is not None:
            torch.cuda.current_stream(self.device).synchronize()  # the old tables may still be in use
            self.seed = int(seed)
            self._create()
            self.state.zero_()
        obs = self._new_obs(out)
        so, do = self._u8(spawn_override, "spawn_override"), self._u8(dir_override, "dir_override")
        _lib.check(self._L.marl_env_reset(self._h, _ptr(self.state), _ptr(obs), _ptr(so), _ptr(do), self._stream()),
                   "marl_env_reset")
        wrapped = self._wrap_obs(obs)
        if self.numpy_io:
            torch.cuda.current_stream(self.device).synchronize()
            if not self.reuse_host_buffers:
                wrapped = {OBS_KEY: wrapped[OBS_KEY].copy()}
        return wrapped, {}

    def step(self, actions, out=None, prio_override=None):
        """agent_environment.py:50."""
        if torch.is_tensor(actions):
            act = actions.to(self.device)
        else:
            act = torch.as_tensor(np.asarray(actions)).to(self.device)
        if act.dtype not in (torch.int64, torch.int32, torch.uint8):
            act = act.to(torch.int64)
        act = act.contiguous()
        if act.numel() != self.N:
            raise ValueError(f"actions must have {self.N} entries, got {tuple(act.shape)}")
        obs = self._new_obs(out)
        rew = torch.empty(self.N, dtype=torch.float32, device=self.device)
        term = torch.empty(self.N, dtype=torch.bool, device=self.device)
        trunc = torch.empty(self.N, dtype=torch.bool, device=self.device)
        po = self._u8(prio_override, "prio_override")
        _lib.check(self._L.marl_env_step(self._h, _ptr(self.state), _ptr(act), act.element_size(), _ptr(obs),
                                         _ptr(rew), _ptr(term), _ptr(trunc), _ptr(po), self._stream()),
                   "marl_env_step")
        if self.numpy_io:
            out = (self._wrap_obs(obs), self._to_host("rew", rew), self._to_host("term", term),
                   self._to_host("trunc", trunc), {})
            self._host_flip ^= 1
            torch.cuda.current_stream(self.device).synchronize()
            if not self.reuse_host_buffers:
                out = ({OBS_KEY: out[0][OBS_KEY].copy()}, out[1].copy(), out[2].copy(), out[3].copy(), out[4])
            return out
        return {OBS_KEY: obs}, rew, term, trunc, {}

    def step_into(self, actions, obs, rewards, terminated, truncated, prio_override=None):
        """Allocation-free step for device-resident loops and graph capture: every argument is
        a preallocated tensor on this env's device (actions int64/int32/uint8 [N]; obs float32
        [N, D]; rewards float32 [N]; terminated/truncated bool or uint8 [N])."""
        # sizes and element widths only (a few hundred nanoseconds): the kernel writes N rows whatever it is handed
        N = self.N
        if (actions.numel() != N or obs.numel() != N * self.D or rewards.numel() != N or terminated.numel() != N
                or truncated.numel() != N or obs.dtype != torch.float32 or rewards.dtype != torch.float32
                or terminated.element_size() != 1 or truncated.element_size() != 1
                or (prio_override is not None and (prio_override.numel() != N or prio_override.element_size() != 1))):
            raise ValueError(f"step_into wants actions [{N}], obs float32 [{N}, {self.D}], rewards float32 [{N}] and one-byte "
                             f"terminated / truncated [{N}] (prio_override uint8 [{N}] if given)")
        _lib.check(self._L.marl_env_step(self._h, _ptr(self.state), _ptr(actions), actions.element_size(), _ptr(obs),
                                         _ptr(rewards), _ptr(terminated), _ptr(truncated), _ptr(prio_override),
                                         self._stream()), "marl_env_step")

    def step_many_into(self, actions, obs, rewards, terminated, truncated, prio_override=None):
        """`actions.shape[0]` consecutive steps in one launch, for action sequences known up front
        (random-action rollouts, replays): actions [K, N]; obs float32 [K, N, D]; rewards float32
        [K, N]; terminated / truncated bool or uint8 [K, N]; all preallocated and contiguous on this
        env's device.  Same results as K `step_into` calls; a step's simulation overlaps the row
        stores of the step before it."""
        K = int(actions.shape[0])
        for name, t, shape in (("actions", actions, (K, self.N)), ("obs", obs, (K, self.N, self.D)),
                               ("rewards", rewards, (K, self.N)), ("terminated", terminated, (K, self.N)),
                               ("truncated", truncated, (K, self.N))):
            if tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"{name} must be a contiguous tensor of shape {shape} on {self.device}")
        if obs.dtype != torch.float32 or rewards.dtype != torch.float32:
            raise ValueError("obs and rewards must be float32")
        if prio_override is not None and (tuple(prio_override.shape) != (K, self.N) or prio_override.dtype != torch.uint8):
            raise ValueError(f"prio_override must be uint8 of shape {(K, self.N)}")
        _lib.check(self._L.marl_env_step_many(self._h, _ptr(self.state), _ptr(actions), actions.element_size(), K,
                                              _ptr(obs), _ptr(rewards), _ptr(terminated), _ptr(truncated),
                                              _ptr(prio_override), self._stream()), "marl_env_step_many")

    def featurize(self, out=None):
        obs = self._new_obs(out)
        _lib.check(self._L.marl_env_featurize(self._h, _ptr(self.state), _ptr(obs), self._stream()),
                   "marl_env_featurize")
        return obs

    # ------------------------------------------------------------------
    def export_state(self):
        """Canonical snapshot as CPU numpy arrays (see include/marl_amd.h)."""
        agents = torch.empty((self.E, self.A, 4), dtype=torch.int32, device=self.device)
        pots = torch.empty((self.E, max(self.NP, 1), 2), dtype=torch.int32, device=self.device)
        items = torch.empty((self.E, self.H * self.W), dtype=torch.uint8, device=self.device)
        misc = torch.empty((self.E, 2), dtype=torch.int32, device=self.device)
        pots_arg = pots[:, : self.NP].contiguous() if self.NP else pots
        _lib.check(self._L.marl_env_export_state(self._h, _ptr(self.state), _ptr(agents), _ptr(pots_arg), _ptr(items),
                                                 _ptr(misc), self._stream()), "marl_env_export_state")
        return {"agents": agents.cpu().numpy(), "pots": pots_arg[:, : self.NP].cpu().numpy(),
                "items": items.cpu().numpy(), "misc": misc.cpu().numpy()}

    def import_state(self, snap):
        agents = np.ascontiguousarray(snap["agents"], np.int32).reshape(self.E, self.A, 4)
        pots = np.ascontiguousarray(snap["pots"], np.int32).reshape(self.E, self.NP, 2)
        items = np.ascontiguousarray(snap["items"], np.uint8).reshape(self.E, self.H * self.W)
        misc = np.ascontiguousarray(snap["misc"], np.int32).reshape(self.E, 2)
        self._validate_snapshot(agents, pots, items, misc)
        d = [torch.from_numpy(x).to(self.device) for x in (agents, pots if self.NP else np.zeros((1,), np.int32), items, misc)]
        _lib.check(self._L.marl_env_import_state(self._h, _ptr(self.state), _ptr(d[0]), _ptr(d[1]), _ptr(d[2]),
                                                 _ptr(d[3]), self._stream()), "marl_env_import_state")
        torch.cuda.current_stream(self.device).synchronize()  # host staging buffers die with this frame

    def _validate_snapshot(self, agents, pots, items, misc):
        grid = np.array([list(r) for r in self.layout_rows])
        floor = np.isin(grid, [" ", "+"]).reshape(-1)
        holder = floor | (grid == "C").reshape(-1)
        cell = agents[..., 0] * self.W + agents[..., 1]
        if (agents[..., 0] < 0).any() or (agents[..., 0] >= self.H).any() or (agents[..., 1] < 0).any() \
                or (agents[..., 1] >= self.W).any() or not floor[cell].all():
            raise ValueError("snapshot puts an agent off the floor")
        if ((agents[..., 2] < 0) | (agents[..., 2] > 3) | (agents[..., 3] < 0) | (agents[..., 3] > 3)).any():
            raise ValueError("snapshot has an out-of-range direction or inventory code")
        if self.A > 1 and (np.sort(cell, axis=1)[:, 1:] == np.sort(cell, axis=1)[:, :-1]).any():
            raise ValueError("snapshot puts two agents on one cell")
        if (items > 3).any() or (items[:, ~holder] != 0).any():
            raise ValueError("snapshot has an item on a cell that cannot hold one")
        if np.take_along_axis(items, cell, axis=1).any():
            raise ValueError("snapshot has an item under an agent")
        if self.NP and ((pots[..., 0] < 0) | (pots[..., 0] > 3) | (pots[..., 1] < 0) | (pots[..., 1] > 30)).any():
            raise ValueError("snapshot has an out-of-range pot")
        if (misc[:, 0] < 0).any():
            raise ValueError("snapshot has a negative step counter")

    # ------------------------------------------------------------------
    def close(self):
        if not self._closed and self._h:
            self._L.marl_env_destroy(self._h)
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

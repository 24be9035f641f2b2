"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's rollout driver,
``agent_environment.agent_environment_loop`` (agent_environment.py:12-130), over host arrays.

PARITY UNPINNED: written from the reference's source as read, statement by statement, and not
checked against an execution of the reference (the reference ships no test or fixture for this
loop).  ``tests/test_oracle_rollout_loop.py`` pins the properties derived from the source by hand
(order of reset/step calls, the one-step-late restart, which row carries which done flag, what the
episode statistics add up).  The product driver (``marl_amd/rollout.py``) is held to this
restatement on the GPU (``tests/test_gpu_rollout_loop.py``).

What the loop does per driver step, with the reference line it follows:
  act on the current observation                                   (:42)
  env.step(actions)                                                (:50)
  reward events of rows 0..1, episode return += mean(rewards[0:2]) (:66-76)
  store (obs, actions, rewards, dones-of-the-PREVIOUS-step, logprobs, values)  (:82-84)
  if all(dones-of-the-previous-step[0:2]): env.reset() replaces next_obs, the episode is booked (:86-104)
  obs = next_obs; dones = terminated | truncated of THIS step      (:106-107)
and ``agent.update(obs)`` after every ``num_steps`` steps (:112).  Because the restart test reads
the flags of the previous step, the driver acts and steps once more after the step that truncated
every env (the vector env has by then restarted them itself): max_steps + 1 env.step calls per episode.
"""
from __future__ import annotations

import numpy as np

OBS_KEY = "n_agent_overcooked_features"


def reference_rollout_loop(agent, env, num_update, num_steps, log=None):
    """agent: act(obs) -> (actions, logprobs, entropy, values[N,1] or None), add_to_buffer(obs, actions,
    rewards, dones, logprobs, values), update(obs).  env: reset() -> (obs dict, info),
    step(actions) -> (obs dict, rewards, terminated, truncated, info), host arrays.
    `log`, if given, receives ("reset",) / ("step",) / ("update",) / ("episode", return) tuples in call order.
    Returns (episode returns, frequency dict) like agent_environment.py:120-130."""
    note = (lambda *a: log.append(a)) if log is not None else (lambda *a: None)
    returns, n_deliver, n_plate, n_pot = [], [], [], []
    ep_return, ep_deliver, ep_plate, ep_pot = 0.0, 0, 0, 0

    first, _ = env.reset()                                          # :34
    note("reset")
    obs = np.asarray(first[OBS_KEY], dtype=np.float32)              # :36
    prev_done = np.zeros(obs.shape[0], dtype=np.float32)            # :37

    for _ in range(num_update):                                     # :40
        for _ in range(num_steps):                                  # :41
            actions, logprobs, _, values = agent.act(obs)           # :42
            nxt, rewards, terminated, truncated, _ = env.step(np.asarray(actions))  # :50
            note("step")
            rewards = np.asarray(rewards, dtype=np.float32)         # :63
            head = rewards[0:2]
            # :66-72, float32 comparisons against Python scalars
            ep_deliver += int(bool(np.any(head >= np.float32(1.0))))
            ep_plate += int(bool(np.any(head == np.float32(0.3))))
            ep_pot += int(bool(np.any(head == np.float32(0.1))))
            ep_return += float(head.mean(dtype=np.float32))         # :76
            if values is not None:
                values = np.asarray(values).reshape(-1)             # :82-83 squeeze(1)
            agent.add_to_buffer(obs, actions, rewards, prev_done, logprobs, values)  # :84
            if np.all(prev_done[0:2] != 0):                         # :86 the flags of the step BEFORE this one
                nxt, _ = env.reset()                                # :88 every env restarts
                note("reset")
                returns.append(ep_return)                           # :89
                note("episode", ep_return)
                n_deliver.append(ep_deliver); n_plate.append(ep_plate); n_pot.append(ep_pot)  # :96-98
                ep_return, ep_deliver, ep_plate, ep_pot = 0.0, 0, 0, 0  # :100-103
            obs = np.asarray(nxt[OBS_KEY], dtype=np.float32)        # :106
            prev_done = np.logical_or(np.asarray(terminated), np.asarray(truncated)).astype(np.float32)  # :107
        agent.update(obs)                                           # :112
        note("update")
    freq = {"frequency_delivery_per_episode": n_deliver, "frequency_plated_per_episode": n_plate,
            "frequency_ingredient_in_pot_per_episode": n_pot}
    return returns, freq


class OracleDictEnv:
    """The CPU oracle env behind the vector-env surface the loop drives (obs dict under OBS_KEY)."""

    def __init__(self, oracle_vec_env):
        self.env = oracle_vec_env

    def reset(self):
        return {OBS_KEY: self.env.reset()}, {}

    def step(self, actions):
        obs, rew, term, trunc = self.env.step(actions)
        return {OBS_KEY: obs}, rew, term, trunc, {}


class ReplayAgent:
    """Plays back recorded actions (one [num_steps, N] array per update) and records what the
    loop stores, one dict of [num_steps, ...] arrays per update."""

    def __init__(self, actions_per_update, logprobs_per_update=None, values_per_update=None):
        self.actions = actions_per_update
        self.logprobs = logprobs_per_update
        self.values = values_per_update
        self.u = 0
        self.t = 0
        self.rows = {k: [] for k in ("obs", "actions", "rewards", "dones", "logprobs", "values")}
        self.snaps = []

    def act(self, obs):
        a = np.asarray(self.actions[self.u][self.t])
        n = a.shape[0]
        lp = np.zeros(n, np.float32) if self.logprobs is None else np.asarray(self.logprobs[self.u][self.t])
        v = np.zeros((n, 1), np.float32) if self.values is None else np.asarray(self.values[self.u][self.t]).reshape(n, 1)
        return a, lp, None, v

    def add_to_buffer(self, obs, actions, rewards, dones, logprobs, values):
        for k, x in zip(("obs", "actions", "rewards", "dones", "logprobs", "values"),
                        (obs, actions, rewards, dones, logprobs, values)):
            self.rows[k].append(np.array(x, dtype=np.float32))  # buffer.py:11-16: every field is float32
        self.t += 1

    def update(self, next_obs):
        snap = {k: np.stack(v) for k, v in self.rows.items()}
        snap["next_obs"] = np.array(next_obs, dtype=np.float32)
        self.snaps.append(snap)
        self.rows = {k: [] for k in self.rows}
        self.u += 1
        self.t = 0

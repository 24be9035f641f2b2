"""Known answers for the CPU restatement of the reference's rollout driver
(oracle/rollout_loop.py, after agent_environment.py:34-112), derived by hand from the source:
the restart test reads the done flags of the PREVIOUS step (:84-88 against :107), so the driver
steps once more on the observation the vector env produced when it restarted itself, stores that
row with dones = 1, books its reward to the old episode, and only then resets every env."""
import numpy as np

from oracle.rollout_loop import OBS_KEY, ReplayAgent, reference_rollout_loop


class ScriptedEnv:
    """N rows in lockstep; truncates on its max_steps-th step and restarts itself inside step()
    like the vector wrapper.  Observation = (driver resets so far, self-restarts so far, t, row)."""

    def __init__(self, n_rows, max_steps, reward_of_step):
        self.n, self.max_steps, self.reward_of_step = n_rows, max_steps, reward_of_step
        self.driver_resets = self.self_restarts = self.t = self.calls = 0

    def _obs(self):
        o = np.empty((self.n, 4), np.float32)
        o[:, 0], o[:, 1], o[:, 2], o[:, 3] = self.driver_resets, self.self_restarts, self.t, np.arange(self.n)
        return {OBS_KEY: o}

    def reset(self):
        self.driver_resets += 1
        self.t = 0
        return self._obs(), {}

    def step(self, actions):
        assert np.asarray(actions).shape == (self.n,)
        self.calls += 1
        self.t += 1
        rew = np.asarray(self.reward_of_step(self.calls), np.float32) * np.ones(self.n, np.float32)
        trunc = np.full(self.n, self.t >= self.max_steps)
        if trunc[0]:
            self.self_restarts += 1
            self.t = 0
        return self._obs(), rew, np.zeros(self.n, bool), trunc, {}


def _run(max_steps=3, T=5, updates=2, reward_of_step=lambda k: 0.0):
    env = ScriptedEnv(4, max_steps, reward_of_step)
    acts = [np.zeros((T, 4), np.int64) for _ in range(updates)]
    agent = ReplayAgent(acts)
    log = []
    returns, freq = reference_rollout_loop(agent, env, updates, T, log=log)
    return env, agent, log, returns, freq


def test_call_order_has_one_extra_step_before_the_driver_reset():
    env, agent, log, returns, _ = _run()
    kinds = [x[0] for x in log]
    assert kinds == ["reset", "step", "step", "step", "step", "reset", "episode", "step", "update",
                     "step", "step", "step", "reset", "episode", "step", "step", "update"]
    assert env.calls == 10 and env.driver_resets == 3 and env.self_restarts == 2
    assert len(returns) == 2


def test_rows_carry_the_previous_steps_flag_and_the_self_restart_observation():
    _, agent, _, _, _ = _run()
    dones = np.concatenate([s["dones"][:, 0] for s in agent.snaps])
    np.testing.assert_array_equal(dones, [0, 0, 0, 1, 0, 0, 0, 1, 0, 0])
    obs = np.concatenate([s["obs"][:, 0, :3] for s in agent.snaps])  # (driver resets, self restarts, t) of row 0
    expect = [(1, 0, 0), (1, 0, 1), (1, 0, 2), (1, 1, 0),  # 4th row: what the env showed after restarting itself
              (2, 1, 0), (2, 1, 1), (2, 1, 2), (2, 2, 0),
              (3, 2, 0), (3, 2, 1)]
    np.testing.assert_array_equal(obs, np.array(expect, np.float32))
    np.testing.assert_array_equal(agent.snaps[0]["next_obs"][0, :3], [2, 1, 1])
    np.testing.assert_array_equal(agent.snaps[1]["next_obs"][0, :3], [3, 2, 2])
    assert agent.snaps[0]["obs"].shape == (5, 4, 4) and agent.snaps[0]["actions"].dtype == np.float32


def test_episode_statistics_include_the_extra_step():
    # step 3 (the truncating one) delivers, step 4 (the extra one) pots an onion, step 5 belongs to episode 2
    rew = {3: 1.0, 4: 0.1, 5: 0.3, 8: 0.3}
    _, _, _, returns, freq = _run(reward_of_step=lambda k: rew.get(k, 0.0))
    f32 = lambda x: float(np.float32(x))
    assert returns == [f32(1.0) + f32(0.1), f32(0.3) + f32(0.3)]
    assert freq["frequency_delivery_per_episode"] == [1, 0]
    assert freq["frequency_ingredient_in_pot_per_episode"] == [1, 0]
    assert freq["frequency_plated_per_episode"] == [0, 2]


def test_oracle_env_under_the_loop_restarts_and_resets(oracle_mod):
    """The same properties on the real CPU env: every episode costs max_steps + 1 steps and the
    row after a truncation holds the restarted env's first observation with dones = 1."""
    from marl_amd.layouts import get_layout
    from oracle.rollout_loop import OracleDictEnv

    E, A, T, updates, max_steps = 3, 2, 7, 3, 4
    ora = oracle_mod.OracleVecEnv(get_layout("overcooked_cramped_room_v0"), A, "global_obs", max_steps=max_steps,
                                  num_envs=E, seed=5)
    rng = np.random.default_rng(0)
    acts = [rng.integers(0, 7, size=(T, E * A)) for _ in range(updates)]
    agent = ReplayAgent(acts)
    log = []
    returns, _ = reference_rollout_loop(agent, OracleDictEnv(ora), updates, T, log=log)
    steps_between_resets = []
    n = 0
    for x in log:
        if x[0] == "step":
            n += 1
        elif x[0] == "reset" and n:
            steps_between_resets.append(n); n = 0
    assert steps_between_resets == [max_steps + 1] * (T * updates // (max_steps + 1))
    dones = np.concatenate([s["dones"][:, 0] for s in agent.snaps])
    assert list(np.nonzero(dones)[0]) == [k for k in range(T * updates) if k % (max_steps + 1) == max_steps]
    assert len(returns) == T * updates // (max_steps + 1)

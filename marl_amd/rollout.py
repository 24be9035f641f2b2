"""Device-resident rollout driver, counterpart of the reference's
``agent_environment.agent_environment_loop`` (agent_environment.py:12-130).

Same control flow -- act, step, count reward events, store, global reset one step after env 0
finishes, update every ``num_steps`` -- with the per-step host round trips removed:
  * observations, rewards and done flags never leave the GPU (the reference does two
    host-to-device copies, one device-to-host copy and a Python loop over rows per step,
    agent_environment.py:50,63,106-107);
  * the env kernel writes the next observation straight into the rollout store's next row;
  * episode statistics accumulate on the device and are read back once per episode;
  * when a rollout is launch-bound (few rows per launch) its steps are captured once into
    hipGraphs and replayed: a rollout without an episode end is one graph; the driver's restart
    at an episode end cuts a rollout in two, and the pieces are replayed from graphs of aligned
    power-of-two runs of steps (a bounded set, whatever the episode length).
The restart keeps the reference's one-step lag (agent_environment.py:84-88,107): the driver tests
the done flags of the PREVIOUS step, so after the step on which every env truncates (and the
vector env has restarted them by itself) it acts and steps once more -- that row is stored with
dones = 1, its rewards count towards the old episode -- and only then resets all envs.  An
episode therefore spans max_steps + 1 driver steps and env.step calls.
Episode statistics keep the reference's definition (agent_environment.py:66-76,86-104: rows
0..1 only, whatever the agent count; a delivery is any reward >= 1, a plating == 0.3, an
onion-in-pot == 0.1; the return is the sum over steps of the mean reward of those rows) and add
the same quantities averaged over every env of every rank; both come from one small kernel
pass (`episode_stats.EpisodeStats`): per step, or, when the rewards go straight into the rollout
store, once per run of steps over the stored rows.
"""
from __future__ import annotations

import torch

from . import dist_utils
from .env import OBS_KEY
from .episode_stats import EpisodeStats


def aligned_runs(first, count):
    """[first, first + count) cut into runs (start, length) whose length is a power of two and
    whose start is a multiple of it: at most 2 log2(count) runs, and over all possible segments
    of a T-step rollout fewer than 2 T distinct ones."""
    out = []
    while count > 0:
        size = 1
        while size * 2 <= count and first % (size * 2) == 0:
            size *= 2
        out.append((first, size))
        first += size
        count -= size
    return out


class ScalarLog:
    """Stand-in for tensorboard's SummaryWriter when tensorboard is not installed: keeps
    scalars in memory (``.scalars[tag] = [(step, value), ...]``) and, when it has a ``log_dir``,
    writes them to ``<log_dir>/scalars.csv`` (tag, step, value) on ``close()``."""

    def __init__(self, log_dir=None):
        self.log_dir = log_dir
        self.scalars = {}

    def add_scalar(self, tag, value, step):
        self.scalars.setdefault(tag, []).append((int(step), float(value)))

    def close(self):
        if not self.log_dir or not self.scalars:
            return
        import csv
        import os

        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, "scalars.csv"), "w", newline="") as f:
            out = csv.writer(f)
            out.writerow(["tag", "step", "value"])
            for tag in sorted(self.scalars):
                for step, value in self.scalars[tag]:
                    out.writerow([tag, step, repr(value)])


def make_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter  # optional dependency

        return SummaryWriter(log_dir=log_dir)
    except Exception:
        return ScalarLog(log_dir)


def agent_environment_loop(agent, env, device, num_update=1000, log_dir=None, args=None, writer=None,
                           progress=False):
    collect_steps = args.num_steps
    A = agent.num_agents
    N = args.num_envs * A
    E = args.num_envs
    if env.num_envs != N:
        raise ValueError(f"env has {env.num_envs} rows, expected num_envs*num_agents = {N}")
    log = bool(getattr(args, "log", False))
    if log and writer is None:
        writer = make_writer(log_dir)
    max_steps = getattr(env, "max_steps", 1000)
    episode_span = max_steps + 1  # driver steps per episode: the reference notices an episode's end one step late
    # "auto": replay the rollout from hipGraphs when it is launch-bound (few rows per launch)
    graph_mode = getattr(args, "graph_rollout", "off")
    # (collectives are issued between graph replays, never captured)
    use_graphs = graph_mode == "on" or (graph_mode == "auto" and N <= 16384)

    episodes_reward, deliveries, platings, pottings = [], [], [], []
    all_env = {"episode_return_all_envs": [], "deliveries_all_envs": [], "platings_all_envs": [],
               "pottings_all_envs": []}
    stats = EpisodeStats(E, A, device)
    steps_in_episode = 0
    num_episodes = 0

    buf = agent.buffer
    dones = torch.zeros(N, dtype=torch.float32, device=device)  # flag produced by the step BEFORE the stored row
    carry = torch.empty_like(buf.obs_slot(0))  # holds the observation that follows the last stored row
    # The env kernel writes rows at 16-byte aligned addresses.  Row t of the rollout store starts
    # t * N * D floats in: aligned for every t when N * D is a multiple of four (always with two or
    # four agents); otherwise (one or three agents with an env count that is no multiple of four) the
    # observations go through two alternating tensors of their own and are copied into the store.
    in_place = buf.obs_slot(0).numel() % 4 == 0
    staging = None if in_place else [torch.empty_like(carry), torch.empty_like(carry)]

    def slot(step):
        if step >= collect_steps:
            return carry
        return buf.obs_slot(step) if in_place else staging[step % 2]

    env.reset(out=slot(0))

    # With the decentralised critic and the fused sampler every producer writes straight into the
    # rollout store: the env kernel the next observation and the rewards, the sampler the
    # log-probabilities, the critic head the values.  What the store wants in another type (actions
    # as float32, the done flag as float32 one row on) is kept per step in its native type and
    # converted once per rollout, and the sampler's call counter moves once per rollout: nine
    # launches fewer per step, a third of a launch-bound step.
    direct = bool(getattr(agent, "can_act_into", lambda: False)()) and buf.max_size == collect_steps and in_place
    if direct:
        act_rows = torch.empty((collect_steps, N), dtype=torch.int64, device=device)
        term_rows = torch.empty((collect_steps, N), dtype=torch.bool, device=device)
        trunc_rows = torch.empty((collect_steps, N), dtype=torch.bool, device=device)
        agent.begin_direct_rollouts(collect_steps)

    # An env made of shards on streams of their own (`sharded_env.ShardedVecEnv`) and a learner whose
    # acting needs nothing beyond a row's own observation: every shard runs act -> step on its own
    # stream for a whole run of steps, with one fork before and one join after the run.  One shard's
    # policy GEMMs then overlap the other's env kernel and the drain of its rows; the stored rows
    # are those of the unsharded run (same env and sampler RNG keys per row).
    pipelined = direct and getattr(env, "num_shards", 1) > 1 \
        and bool(getattr(agent, "can_act_on_shards", lambda: False)())

    def shard_step(k, shard, rows, step):
        agent.act_into(slot(step)[rows], step, act_rows[step][rows], rows=rows)
        shard.step_into(act_rows[step][rows], slot(step + 1)[rows], buf.rewards_buff[step][rows], term_rows[step][rows],
                        trunc_rows[step][rows])

    def run_steps(first, count):
        if pipelined:
            env.run_pipelined(range(first, first + count), shard_step)
            buf.ptr = (first + count) % buf.max_size
            buf.size = min(max(buf.size, first + count), buf.max_size)
        else:
            for step in range(first, first + count):
                one_step(step)

    def one_step(step):
        # every tensor touched here has a fixed address per step index, so the same code can be
        # captured into a graph and replayed
        obs = slot(step)
        if direct:
            agent.act_into(obs, step, act_rows[step])
            env.step_into(act_rows[step], slot(step + 1), buf.rewards_buff[step], term_rows[step], trunc_rows[step])
            # (the reward events of these steps are counted per segment, from the stored rows: run_segment)
            buf.ptr = (step + 1) % buf.max_size
            buf.size = min(max(buf.size, step + 1), buf.max_size)
            return
        actions, logprobs, _, values = agent.act(obs)
        _, rewards, term, trunc, _ = env.step(actions, out=slot(step + 1))
        stats.add(rewards)
        agent.add_to_buffer(obs, actions, rewards, dones, logprobs, values.squeeze(1))
        dones.copy_(term | trunc)

    graphs = {}
    pool = None

    def replay_run(first, count):
        nonlocal pool
        g = graphs.get((first, count))
        if g is None:
            # capture records the launches without running them; the replay below does the work
            buf.ptr = first
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                run_steps(first, count)
            if pool is None:
                pool = g.pool()
            graphs[(first, count)] = g
        g.replay()

    def run_segment(first, count, warm):
        """steps [first, first + count) of the current rollout"""
        if not (use_graphs and warm):
            run_steps(first, count)
            if direct:
                stats.add_many(buf.rewards_buff[first:first + count])
            return
        if count == collect_steps:
            replay_run(first, count)  # a whole rollout: one graph
        else:
            for f, c in aligned_runs(first, count):
                replay_run(f, c)
        if direct:
            stats.add_many(buf.rewards_buff[first:first + count])
        buf.ptr = (first + count) % buf.max_size
        buf.size = min(first + count, buf.max_size)

    rng = range(num_update)
    if progress:
        try:
            from tqdm import tqdm

            rng = tqdm(rng)
        except Exception:
            pass
    for update_index in rng:
        step = 0
        if direct:
            agent.begin_direct_rollout()
        while step < collect_steps:
            count = min(collect_steps - step, episode_span - steps_in_episode)
            run_segment(step, count, warm=update_index > 0)  # the first rollout runs eagerly (lazy library set-up)
            step += count
            steps_in_episode += count
            if steps_in_episode >= episode_span:
                # All envs run in lockstep: they truncated (and restarted themselves) on driver step
                # max_steps of this episode, and the step just done is the one the reference makes
                # before it notices (its test reads the previous step's flags,
                # agent_environment.py:84-88,107).  Now it restarts every env from the driver; the
                # reset observation replaces the one that step returned.
                env.reset(out=slot(step))
                totals, first_rows = stats.raw()
                tot_all = totals.to(torch.float64)
                if dist_utils.collectives_active():
                    dist_utils.all_reduce_sum_(tot_all)
                    tot_all /= dist_utils.world_size()
                t = tot_all.tolist()  # the one host sync per episode
                vals = first_rows.tolist() + [t[0] / 10.0 / N, t[1] / E, t[2] / E, t[3] / E]
                episodes_reward.append(vals[0]); deliveries.append(vals[1]); platings.append(vals[2]); pottings.append(vals[3])
                for k, v in zip(all_env, vals[4:]):
                    all_env[k].append(v)
                if log and writer is not None:
                    writer.add_scalar("episode_rewards", vals[0], num_episodes)
                    writer.add_scalar("freq/frequency_delivery_per_episode", vals[1], num_episodes)
                    writer.add_scalar("freq/frequency_plated_per_episode", vals[2], num_episodes)
                    writer.add_scalar("freq/frequency_ingredient_in_pot_per_episode", vals[3], num_episodes)
                    writer.add_scalar("all_envs/episode_return", vals[4], num_episodes)
                    writer.add_scalar("all_envs/deliveries_per_episode", vals[5], num_episodes)
                stats.reset()
                steps_in_episode = 0
                num_episodes += 1
        if direct:
            # row k of the store holds the flag of the step BEFORE it: rows 1.. from this rollout's
            # steps, row 0 from the previous rollout's last step (copied below, after the update)
            if collect_steps > 1:
                torch.logical_or(term_rows[:-1], trunc_rows[:-1], out=buf.dones_buff[1:])
            torch.logical_or(term_rows[-1], trunc_rows[-1], out=dones)
            agent.end_direct_rollout(act_rows)
        agent.update(carry)
        # the stored rows restart at 0: the carried observation (and its done flag) becomes row 0
        slot(0).copy_(carry)
        if direct:
            buf.dones_buff[0].copy_(dones)

    freq = {
        "frequency_delivery_per_episode": deliveries,
        "frequency_plated_per_episode": platings,
        "frequency_ingredient_in_pot_per_episode": pottings,
    }
    freq.update(all_env)
    return episodes_reward, freq

#!/usr/bin/env python
"""Headline benchmark: batched Overcooked step + featurize throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one vectorised env.step (dynamics + featurize + auto-reset, one kernel launch) over
the per-GPU batch of BASELINE.json's configs[1]: overcooked_cramped_room_v0, 2 agents, 65,536
envs, global_obs, uniform random actions (pre-generated on the device).  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank steps its own 65,536 envs (weak scaling; env
RNG keys are offset by rank so the job equals one big env batch); there is no data-path
collective.  Rank 0 prints one JSON line.

Before the W warm-up steps the process steps the envs untimed for --spinup-ms (300 ms; clocks).
What is timed, and what the extra objects in the line mean:
  * the timed K steps write every step's observation rows into their own tensor of a ring of 32
    (3.4 GB), as a rollout store keeps them: the rows have to drain to HBM, and `value`, `roofline`
    are this HBM-bound regime;
  * `llc_resident`: the same launches alternating between TWO observation tensors (212 MB), which
    stay in the 256 MiB last-level cache and are overwritten before they ever reach HBM: faster,
    but a cache number, labelled as such;
  * `step_many`: the multi-step launch (marl_env_step_many, 32 steps per launch, usable when the
    actions are known up front as they are here), rows kept in the same ring: HBM regime;
  * `two_shards`: the batch as two half-batch env handles on two streams, single-step launches
    replayed from one hipGraph without a join between steps (one shard simulates while the other
    streams its rows), rows kept: HBM regime; `cached_actions` inside it: the same with the actions
    read from the L2s instead of HBM;
  * `policy_driven`: actions sampled every step from the reference-shaped fp32 MLP (every rank, at
    any N);
  * `peak_measured`: what a plain device fill / copy of the same ring reaches in this run;
  * `cpu_baseline`: the CPU restatement (oracle) on the host's cores, a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def algorithmic_bytes_per_agent_step(D, state_words, A, action_bytes):
    # obs write + reward + terminated + truncated + action read + state read and write
    return 4 * D + 4 + 1 + 1 + action_bytes + 2 * (4 * state_words) / A


def host_threads():
    """Threads for the CPU baseline: every host core this process may use.  os.cpu_count() is the
    whole machine; a GPU box hands each job a share of it (scheduler affinity and the
    OMP_NUM_THREADS it exports), and more threads than that share only time-slice."""
    total = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = total
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = min(total, affinity, share) if share > 0 else min(total, affinity)
    return max(threads, 1), total, affinity, share


def cpu_baseline(layout_rows, A, feature, E, seed, single_thread=False, budget_s=15.0):
    """Time the CPU oracle (plain-C restatement, OpenMP over envs) on a bounded sample of the
    same workload: on every core this job may use, or (single_thread) on one.  Reported as a
    baseline only."""
    import numpy as np

    import oracle

    threads, total, affinity, share = host_threads()
    if single_thread:
        threads, E = 1, min(E, 4096)  # one thread needs no more envs than keep it busy for a step
    oracle.build()
    env = oracle.OracleVecEnv(layout_rows, A, feature, max_steps=1000, num_envs=E, seed=seed, nthreads=threads)
    env.reset()
    rng = np.random.default_rng(seed)
    acts = rng.integers(0, 7, size=(8, E * A)).astype(np.int64)
    t0 = time.perf_counter()
    for i in range(4):
        env.step(acts[i % 8])
    per = (time.perf_counter() - t0) / 4
    steps = int(max(8, min(6000, budget_s / max(per, 1e-6))))  # about budget_s seconds of host work
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i % 8])
    dt = time.perf_counter() - t0
    return {
        "value": E * A * steps / dt,
        "unit": "agent-steps/s",
        "cores": threads,
        "kind": "port",
        "host": {"os_cpu_count": total, "sched_affinity": affinity, "OMP_NUM_THREADS": share or None},
        "sample": f"{steps} steps x {E} envs x {A} agents, same layout/features/actions distribution, "
                  f"CPU restatement (not cogrid) on {threads} OpenMP thread(s)"
                  + ("" if single_thread else " = every core this job may use") + f", {dt:.1f} s",
    }


def free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def rank_launch_command(n, argv, port):
    """The command `python bench.py --gpus N ...` turns itself into: one rank per GPU under
    torch.distributed.run, rendezvous on the loopback address."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(n, argv):
    """Run the N ranks as a child process tree and hand back its exit code.  Called before anything
    in this process has initialised the GPU; the child's stdout (rank 0's JSON line) and stderr pass
    straight through."""
    import subprocess

    # (counting devices does not initialise the GPU)
    if os.environ.get("MARL_AMD_DIST_BACKEND", "nccl") == "nccl" and n > torch.cuda.device_count():
        print(f"bench.py: --gpus {n} but {torch.cuda.device_count()} GPU(s) are visible: one rank per GPU over RCCL needs {n} "
              "(MARL_AMD_DIST_BACKEND=gloo lets ranks share a GPU, as a rehearsal of the code path only)", file=sys.stderr)
        return 2
    return subprocess.run(rank_launch_command(n, argv, free_port())).returncode


def timed(fn, count, dev):
    """`count` calls of fn(i) between two HIP events on the current (launch) stream; us per call"""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    ev0.record()
    for i in range(count):
        fn(i)
    ev1.record()
    torch.cuda.synchronize(dev)
    return ev0.elapsed_time(ev1) * 1e3 / count


def policy_leg(env, store, rew, term, trunc, N, dev, rank, A, world, barrier, reduce_max, shards=1):
    """Policy-driven rollout on every rank (MLP forward + sample + step, the policy reading what the
    step wrote), bracketed like the timed region: barrier + synchronize on both sides, the slowest
    rank's time counts."""
    from marl_amd.policy import Agent

    ring = store.shape[0]
    torch.manual_seed(1)
    policy = Agent(env.single_observation_space, env.single_action_space, num_agents=A, num_envs=1,
                   build_centralised=False).to(dev)  # the centralised critic is not evaluated on this path
    policy.use_fused_sampling(1234, row_index_base=rank * N)  # sample + log_prob + entropy in one launch
    if shards > 1:
        # the product's pipelined form (as rollout.agent_environment_loop drives a sharded env): every
        # shard runs forward -> sample -> step on its own stream, no join between steps
        from marl_amd.sharded_env import ShardedVecEnv

        env = ShardedVecEnv(env.layout_rows, A, env.feature, env.E, shards=shards, max_steps=env.max_steps, seed=1234,
                            env_index_base=rank * env.E, device=dev)
    env.reset(out=store[0])

    # the rollout half of act() as the trainer's rollouts run it (`Agent.rollout_step`): both heads from
    # one GEMM, actions, log-probabilities and values written in place
    act = torch.empty(N, dtype=torch.int64, device=dev)
    logp = torch.empty(N, dtype=torch.float32, device=dev)
    val = torch.empty(N, dtype=torch.float32, device=dev)

    def pstep(i):
        policy.rollout_step(store[i % ring], logp, val, actions_out=act)
        env.step_into(act, store[(i + 1) % ring], rew, term, trunc)

    def shard_step(k, shard, rows, i):
        policy.rollout_step(store[i % ring][rows], logp[rows], val[rows], actions_out=act[rows], row_offset=rows.start)
        shard.step_into(act[rows], store[(i + 1) % ring][rows], rew[rows], term[rows], trunc[rows])

    def run(first, count):
        if shards > 1:
            env.run_pipelined(range(first, first + count), shard_step)
        else:
            for i in range(first, first + count):
                pstep(i)

    steps = 128
    run(0, 16)
    barrier()
    t0 = time.perf_counter()
    run(16, steps)
    barrier()
    wall = reduce_max(time.perf_counter() - t0)
    env_us = None
    if shards == 1:
        # the env launch as it runs inside this loop (a policy forward between two steps), outside the timed
        # steps: an event on either side of it, median over 64 steps
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        for j, (e0, e1) in enumerate(evs):
            i = 16 + steps + j
            policy.rollout_step(store[i % ring], logp, val, actions_out=act)
            e0.record()
            env.step_into(act, store[(i + 1) % ring], rew, term, trunc)
            e1.record()
        torch.cuda.synchronize(dev)
        env_us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)[len(evs) // 2]
    return {
        "metric": "env agent-steps/sec (policy-driven rollout: fp32 MLP forward + sample + fused step+featurize)",
        "value": world * N * steps / wall, "unit": "agent-steps/s", "n_gpus": world, "ms_per_step": wall / steps * 1e3,
        "steps": steps, "env_launch_us_in_loop": env_us,
        "env_launch_us_in_loop_note": None if env_us is None else
        "event to event around the env launch between two policy forwards, median of 64; compare launch_time_us.median "
        "(the same measure for back-to-back launches; the events add ~3 us to both)",
        "policy": "obs->512->256->7/1 fp32 MLP (PyTorch-ROCm GEMMs, both heads in one), fused categorical sampling; rows kept; "
                  "every rank its own envs, no collective, slowest rank's wall time"
                  + (f"; the envs of a rank as {shards} shards on streams of their own, each running forward -> sample -> "
                     "step without a join between steps (ShardedVecEnv.run_pipelined)" if shards > 1 else ""),
        "shards": shards}


def two_shard_leg(env, actions, store, N, D, B, dev, rank, A, E):
    """The product's sharded env (marl_amd.sharded_env.ShardedVecEnv) in its pipelined form: every
    shard steps on its own stream, replayed from one hipGraph with a branch per shard."""
    from marl_amd.sharded_env import ShardedVecEnv

    ring = store.shape[0]
    sh = ShardedVecEnv(env.layout_rows, A, env.feature, E, shards=2, max_steps=env.max_steps, seed=1234,
                       env_index_base=rank * E, device=dev)
    rew = torch.empty(N, dtype=torch.float32, device=dev)
    term = torch.empty(N, dtype=torch.bool, device=dev)
    trunc = torch.empty(N, dtype=torch.bool, device=dev)
    sh.reset(out=store[0])

    def measure(batches):
        def body(k, shard, rows, i):
            shard.step_into(actions[i % batches][rows], store[(i + 1) % ring][rows], rew[rows], term[rows], trunc[rows])

        sh.run_pipelined(range(ring), body)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sh.run_pipelined(range(ring), body)
        g.replay()
        reps = 16
        return timed(lambda i: g.replay(), reps, dev) / ring

    R = actions.shape[0]
    us = measure(R)
    # the same with at most 8 action batches in rotation: they stay in the XCD L2s between uses, as actions a
    # sampler has just written do, instead of being fetched from HBM behind the other shard's row stream
    us_cached = measure(min(R, 8))
    return {"regime": f"marl_amd.sharded_env.ShardedVecEnv, two shards of {E // 2} envs on two streams, single-step launches, "
                      f"run_pipelined over {ring} steps captured into one hipGraph (a branch per shard, no join between "
                      "steps); every step's rows kept: HBM",
            "bound": "hbm", "us_per_step": us, "value": N / us * 1e6, "achieved": N * B / us / 1e3, "unit": "GB/s",
            "frac": N * B / us / 1e3 / HBM_PEAK_GBS,
            "cached_actions": {"action_batches": min(R, 8), "us_per_step": us_cached, "value": N / us_cached * 1e6,
                               "frac": N * B / us_cached / 1e3 / HBM_PEAK_GBS,
                               "what": "the same graph with the steps' actions read from the L2s (few batches in rotation) "
                                       "instead of HBM: what a per-shard sampler's actions cost to read"}}


def guarded(out, name, fn):
    """An extra leg that raises must not take the headline with it: its entry then carries the error."""
    try:
        val = fn()
        if val is not None:
            out[name] = val
    except Exception as ex:  # noqa: BLE001
        import traceback

        traceback.print_exc(file=sys.stderr)
        out[name] = {"error": f"{type(ex).__name__}: {ex}"}


def extra_legs(env, actions, store, rew, term, trunc, N, D, B, dev, rank, A, E):
    """What rank 0 adds beyond the timed region on a single GPU."""
    out = {}
    R, ring = actions.shape[0], store.shape[0]
    frac = lambda us: N * B / us / 1e3 / HBM_PEAK_GBS  # noqa: E731

    def llc_leg():
        # two alternating tensors: the rows never have to leave the last-level cache
        for i in range(64):
            env.step_into(actions[i % R], store[i % 2], rew, term, trunc)
        us = timed(lambda i: env.step_into(actions[i % R], store[i % 2], rew, term, trunc), 1024, dev)
        return {
            "regime": (f"obs_ring=2: {2 * N * D * 4 / 1e6:.0f} MB of rows stay in the 256 MiB last-level cache and are overwritten "
                       "there; cache-resident store rate, NOT HBM bandwidth") if 2 * N * D * 4 <= (256 << 20) else
                      f"obs_ring=2: {2 * N * D * 4 / 1e6:.0f} MB of rows, more than the 256 MiB last-level cache holds: they drain to HBM",
            "bound": "llc" if 2 * N * D * 4 <= (256 << 20) else "hbm", "us_per_launch": us, "value": N / us * 1e6,
            "achieved": N * B / us / 1e3, "unit": "GB/s", "frac_of_hbm_spec": frac(us)}

    def step_many_leg():
        # the multi-step launch, rows kept
        if (N * D) % 4 != 0:
            return None
        rew_k = torch.empty((ring, N), dtype=torch.float32, device=dev)
        te_k = torch.empty((ring, N), dtype=torch.bool, device=dev)
        tr_k = torch.empty_like(te_k)
        acts_k = actions[torch.arange(ring, device=dev) % R].contiguous()  # one [ring, N] array: the launch prefetches a step ahead
        env.step_many_into(acts_k, store, rew_k, te_k, tr_k)
        reps = 16
        us_m = timed(lambda i: env.step_many_into(acts_k, store, rew_k, te_k, tr_k), reps, dev) / ring
        return {
            "regime": f"marl_env_step_many, {ring} steps per launch, every step's rows kept (obs_ring={ring}): HBM",
            "bound": "hbm", "steps_per_launch": ring, "us_per_step": us_m, "value": N / us_m * 1e6,
            "achieved": N * B / us_m / 1e3, "unit": "GB/s", "frac": frac(us_m)}

    def shards_leg():
        # two half-batch shards (env handles of their own) on two streams, single-step launches, rows
        # kept; one hipGraph with a branch per shard and no join between steps, so the shards drift out
        # of phase and one shard's simulation runs while the other streams its rows
        if E % 2 != 0 or ((N // 2) * D) % 4 != 0:
            return None
        return two_shard_leg(env, actions, store, N, D, B, dev, rank, A, E)

    def peak_leg():
        # what plain device writes / copies of the same memory reach in this run
        flat = store.view(-1)
        for _ in range(2):
            flat.fill_(1.0)
        us_fill = timed(lambda i: flat.fill_(float(i)), 8, dev)
        half = flat.numel() // 2
        a, b = flat[:half], flat[half:2 * half]
        b.copy_(a)
        us_copy = timed(lambda i: b.copy_(a), 8, dev)
        return {
            "fill_GBps": flat.numel() * 4 / us_fill / 1e3, "copy_GBps": 2 * half * 4 / us_copy / 1e3,
            "what": f"torch fill_ of the {flat.numel() * 4 / 1e9:.1f} GB ring (write only) and a copy of one half onto the "
                    "other (read + write bytes counted), this run"}

    def launch_dist_leg():
        # distribution of the single-step launch time, rows kept: an event between every two launches
        # (the events cost a little themselves, so the median sits slightly above roofline.us_per_launch)
        n = 1024
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        for i in range(ring):
            env.step_into(actions[i % R], store[(i + 1) % ring], rew, term, trunc)
        torch.cuda.synchronize(dev)
        evs[0].record()
        for i in range(n):
            env.step_into(actions[i % R], store[(i + 1) % ring], rew, term, trunc)
            evs[i + 1].record()
        torch.cuda.synchronize(dev)
        us = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n))
        return {"n": n, "min": us[0], "p10": us[n // 10], "median": us[n // 2], "p90": us[n * 9 // 10], "p99": us[n * 99 // 100],
                "max": us[-1], "unit": "us per launch, event to event on the launch stream"}

    guarded(out, "launch_time_us", launch_dist_leg)
    guarded(out, "llc_resident", llc_leg)
    guarded(out, "step_many", step_many_leg)
    guarded(out, "two_shards", shards_leg)
    guarded(out, "peak_measured", peak_leg)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--layout", default="overcooked_cramped_room_v0")
    ap.add_argument("--agents", type=int, default=2)
    ap.add_argument("--feature", default="global_obs")
    ap.add_argument("--config", type=int, choices=[2, 4, 5], default=None,
                    help="preset of a BASELINE.json config: 2 (default workload) cramped_room / 2 agents / 65,536 envs / global_obs; "
                         "4 counter_circuit / 2 agents / 131,072 envs per GPU (1 M envs over 8 GPUs) / global_obs; "
                         "5 coordination_ring / 4 agents / 65,536 envs / local_obs")
    ap.add_argument("--graph", action="store_true", help="replay the steps from a hipGraph of 32 launches instead of launching eagerly")
    ap.add_argument("--no-graph", action="store_true", help="(default) launch every step eagerly from Python")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", "--no-rows-kept", dest="no_extra_legs", action="store_true",
                    help="only the timed region (profiling runs: keeps the traced launches uniform)")
    ap.add_argument("--obs-ring", type=int, default=32,
                    help="observation tensors the timed steps cycle through (32: every step's rows are kept like in a "
                         "rollout store and drain to HBM; 2: the next step overwrites the rows the consumer has just "
                         "read, they never leave the last-level cache)")
    ap.add_argument("--action-batches", type=int, default=32,
                    help="distinct pre-generated action batches the steps cycle through (int64, 8 bytes per row).  32 batches "
                         "of 131,072 rows are 4 MB per XCD, as much as its L2 holds, so every launch reads its actions from HBM "
                         "(the default: what a caller with a long pre-generated action sequence sees); 16 or fewer stay in the "
                         "L2s between uses, as actions a sampler has just written do")
    ap.add_argument("--spinup-ms", type=float, default=300.0,
                    help="untimed env steps for this long before the warm-up steps, so that a short run does not measure "
                         "the GPU's power-saving state (0 = none)")
    ap.add_argument("--single-gpu-value", type=float, default=None,
                    help="agent-steps/s of the same command at --gpus 1: adds scaling_efficiency = value / (N x this) to the line")
    ap.add_argument("--mode", choices=["random", "policy"], default="random",
                    help="random: pre-generated uniform actions (headline); policy: the timed steps sample their actions "
                         "from the reference-shaped MLP (obs->512->256->7/1, fp32, PyTorch-ROCm)")
    args = ap.parse_args()
    if args.config == 4:
        args.layout, args.agents, args.feature, args.envs = "overcooked_counter_circuit_v0", 2, "global_obs", 131072
    elif args.config == 5:
        args.layout, args.agents, args.feature, args.envs = "overcooked_coordination_ring_v0", 4, "local_obs", 65536

    if args.gpus > 1 and "RANK" not in os.environ:
        # started as `python bench.py --gpus N`: this process becomes the launcher and never touches
        # the GPU; the ranks are fresh children (one per GPU) of torch.distributed.run
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    backend = os.environ.get("MARL_AMD_DIST_BACKEND", "nccl")  # gloo: several ranks on one GPU (rehearsal only)
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run, also with one rank
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL
        else:
            dist.init_process_group(backend)
        group = backend

    from marl_amd.env import BatchedOvercookedVecEnv
    from marl_amd.layouts import get_layout

    E, A = args.envs, args.agents
    rows = get_layout(args.layout)
    env = BatchedOvercookedVecEnv(rows, A, args.feature, E, max_steps=1000, seed=1234, env_index_base=rank * E,
                                  device=dev)
    N, D = env.N, env.D
    if (N * D) % 4:
        raise SystemExit(f"--envs {E} with {A} agents and {args.feature}: the rows of one step ({N} x {D} floats) must be a "
                         "multiple of 16 bytes for the tensors of the ring to start aligned; use an env count that is a multiple of 4")
    R = args.action_batches  # distinct pre-generated action batches, cycled
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    actions = torch.randint(0, 7, (R, N), dtype=torch.int64, device=dev, generator=gen)
    RING = int(max(2, min(args.obs_ring, (8 << 30) // (N * D * 4))))  # at most 8 GB of rows
    store = torch.empty((RING, N, D), dtype=torch.float32, device=dev)
    store.zero_()  # a rollout store is written rollout after rollout: no timed step is the first touch of its rows
    rew = torch.empty(N, dtype=torch.float32, device=dev)
    term = torch.empty(N, dtype=torch.bool, device=dev)
    trunc = torch.empty(N, dtype=torch.bool, device=dev)
    env.reset(out=store[0])

    policy = None
    if args.mode == "policy":
        from marl_amd.policy import Agent

        torch.manual_seed(1)
        policy = Agent(env.single_observation_space, env.single_action_space, num_agents=A, num_envs=1,
                       build_centralised=False).to(dev)
        policy.use_fused_sampling(1234, row_index_base=rank * N)
        p_act = torch.empty(N, dtype=torch.int64, device=dev)
        p_logp = torch.empty(N, dtype=torch.float32, device=dev)
        p_val = torch.empty(N, dtype=torch.float32, device=dev)

    def eager(i):
        # step i reads (policy mode) the rows in slot i and writes slot i + 1; slot 0 holds the reset's rows
        if policy is None:
            env.step_into(actions[i % R], store[(i + 1) % RING], rew, term, trunc)
        else:
            policy.rollout_step(store[i % RING], p_logp, p_val, actions_out=p_act)
            env.step_into(p_act, store[(i + 1) % RING], rew, term, trunc)

    # Device spin-up, before the W warm-up steps and outside every measurement: a run of a few dozen
    # launches (--steps 20 --warmup 5 is 0.6 ms of GPU work) otherwise starts timing on a GPU that has
    # been busy for a millisecond since its context was created.  Measured with those flags: 22.5 us per
    # launch after 300 ms of untimed steps, 22.6-23.2 us without.
    if args.spinup_ms > 0:
        t_end = time.perf_counter() + args.spinup_ms * 1e-3
        i = 0
        while time.perf_counter() < t_end:
            for _ in range(64):
                env.step_into(actions[i % R], store[(i + 1) % RING], rew, term, trunc)
                i += 1
            torch.cuda.synchronize()
        env.reset(out=store[0])
    for i in range(args.warmup):
        eager(i)
    torch.cuda.synchronize()

    graph = None
    GL = 32  # steps per graph
    if args.graph and not args.no_graph and args.steps >= GL and RING in (2, 4, 8, 16, 32):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(GL):
                eager(i)
        graph.replay()  # one untimed replay
        torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    done = 0
    if graph is not None:
        for _ in range(args.steps // GL):
            graph.replay()
        done = (args.steps // GL) * GL
    for i in range(done, args.steps):
        eager(i)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)

    def reduce_max(*vals):
        if group is None:
            return vals if len(vals) > 1 else vals[0]
        tt = torch.tensor(vals, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out = tt.tolist()
        return out if len(out) > 1 else out[0]

    my_dev_ms = dev_ms
    wall, dev_ms = reduce_max(wall, dev_ms)
    ranks_seen, per_rank_us = 1, [my_dev_ms * 1e3 / args.steps]
    if group is not None:
        # proof that the collective library saw every rank: a SUM all-reduce of ones (on the device
        # under RCCL) and the per-rank launch times gathered to everybody
        cdev = dev if backend == "nccl" else "cpu"
        ones = torch.ones(1, dtype=torch.float32, device=cdev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
        mine = torch.tensor([my_dev_ms * 1e3 / args.steps], dtype=torch.float64, device=cdev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank_us = [float(g.item()) for g in gathered]
    # north star: random-action AND policy-driven rates at every GPU count
    policy_line = policy_sharded = None
    if policy is None and not args.no_extra_legs:
        if world == 1:
            holder = {}
            guarded(holder, "policy_driven", lambda: policy_leg(env, store, rew, term, trunc, N, dev, rank, A, world, barrier, reduce_max))
            policy_line = holder.get("policy_driven")
            if E % 2 == 0 and ((N // 2) * D) % 4 == 0:
                guarded(holder, "policy_driven_two_shards",
                        lambda: policy_leg(env, store, rew, term, trunc, N, dev, rank, A, world, barrier, reduce_max, shards=2))
                policy_sharded = holder.get("policy_driven_two_shards")
        else:  # several ranks: a rank that fails must end the job, not leave the others at a barrier
            policy_line = policy_leg(env, store, rew, term, trunc, N, dev, rank, A, world, barrier, reduce_max)

    if rank == 0:
        B = algorithmic_bytes_per_agent_step(D, env.state_words, A, actions.element_size())
        launch_s = dev_ms * 1e-3 / args.steps  # device-side time per launch, HIP events on the launch stream
        achieved = N * B / launch_s / 1e9
        kept = RING * N * D * 4 > (256 << 20)  # more rows alive than the last-level cache holds
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        key = f"{args.layout}/{A}/{args.feature}/{E}/" + ("kept" if kept else "llc")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(key)
        out = {
            "metric": "env agent-steps/sec (fused step+featurize, random actions)" if policy is None
                      else "env agent-steps/sec (policy-driven rollout: MLP forward + sample + fused step+featurize)",
            "value": world * N * args.steps / wall,
            "unit": "agent-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i32 state, f32 observations",
            "data": "synthetic",
            "config": {"workload": f"{args.layout}, {A} agents, {E} envs per GPU, {args.feature} (D={D}), "
                                   + ("uniform random int64 actions" if policy is None else "actions sampled from the fp32 MLP policy")
                                   + ", auto-reset at t=1000, one launch per step, "
                                   + (f"every step's rows kept (ring of {RING} observation tensors, {RING * N * D * 4 / 1e9:.1f} GB)"
                                      if kept else f"{RING} alternating observation tensors (cache resident)"),
                       "launch": "hipGraph of 32 steps" if graph is not None else "eager",
                       "obs_ring": RING, "action_batches": R, "process_group": group, "spinup_ms": args.spinup_ms,
                       "world_size": dist.get_world_size() if group is not None else 1,
                       "all_reduce_ranks": ranks_seen,
                       "rccl_ranks": ranks_seen if group == "nccl" else None,
                       "envs_total": world * E},
            "roofline": {"bound": "hbm" if kept else "llc",
                         "regime": ("rows kept: they drain to HBM while they are written" if kept else
                                    "rows overwritten inside the 256 MiB last-level cache: a cache number, not HBM bandwidth"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": None if traffic is None else
                         "profiles/traffic.json: rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE per launch) of an earlier "
                         "run of this command, not measured in this run",
                         "bytes_per_agent_step": B, "kernel": f"env_kernel<{A} agents, {args.feature}> (step+featurize)",
                         "us_per_launch": launch_s * 1e6,
                         "us_per_launch_per_rank": per_rank_us,
                         "which_rank": "slowest rank (max over ranks of the HIP-event time)" if world > 1 else "rank 0",
                         # by the clock `value` uses (host wall time around the timed region, launch gaps included)
                         "frac_wall": N * B / (wall / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        if args.single_gpu_value:
            out["scaling_efficiency"] = out["value"] / (world * args.single_gpu_value)
        if policy is not None:
            # the per-step time now contains the MLP forward pass: no single-kernel roofline applies
            out["roofline"] = None
        if world == 1 and policy is None and not args.no_extra_legs:
            legs = extra_legs(env, actions, store, rew, term, trunc, N, D, B, dev, rank, A, E)
            pm = legs.pop("peak_measured", None)
            out["roofline"]["peak_measured"] = pm
            if pm and "fill_GBps" in pm:
                out["roofline"]["frac_of_measured_fill"] = achieved / pm["fill_GBps"]
            out.update(legs)
        if policy_line is not None:
            out["policy_driven"] = policy_line
        if policy_sharded is not None:
            out["policy_driven_two_shards"] = policy_sharded
        if world == 1 and not args.no_cpu_baseline:
            guarded(out, "cpu_baseline", lambda: cpu_baseline(rows, A, args.feature, E, 1234))
            guarded(out, "cpu_baseline_1t", lambda: cpu_baseline(rows, A, args.feature, E, 1234, single_thread=True, budget_s=3.0))
        print(json.dumps(out), flush=True)
    if group is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

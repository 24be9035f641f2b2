#!/usr/bin/env python
"""Headline benchmark: batched Overcooked step + featurize throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one vectorised env.step (dynamics + featurize + auto-reset) over the
per-GPU batch of BASELINE.json's configs[1]: overcooked_cramped_room_v0, 2 agents,
65,536 envs, global_obs, uniform random actions (pre-generated on the device).  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank steps its own 65,536
envs (weak scaling; env RNG keys are offset by rank so the job equals one big env batch);
there is no data-path collective.  Rank 0 prints one JSON line.
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


def cpu_baseline(layout_rows, A, feature, E, seed, threads):
    """Time the CPU oracle (plain-C restatement, OpenMP over envs) on a bounded sample of the
    same workload.  Reported as a baseline only."""
    import numpy as np

    import oracle

    oracle.build()
    env = oracle.OracleVecEnv(layout_rows, A, feature, max_steps=1000, num_envs=E, seed=seed, nthreads=threads)
    env.reset()
    rng = np.random.default_rng(seed)
    acts = rng.integers(0, 7, size=(8, E * A)).astype(np.int64)
    t0 = time.perf_counter()
    for i in range(4):
        env.step(acts[i % 8])
    per = (time.perf_counter() - t0) / 4
    steps = int(max(8, min(6000, 15.0 / max(per, 1e-6))))  # about 15 s of host work
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i % 8])
    dt = time.perf_counter() - t0
    return {
        "value": E * A * steps / dt,
        "unit": "agent-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{steps} steps x {E} envs x {A} agents, same layout/features/actions distribution, "
                  f"CPU restatement (not cogrid), {dt:.1f} s",
    }


def rows_kept_leg(env, actions, rew, term, trunc, N, D, B, ring=32, steps=512):
    """The same launches when every step's rows are kept (a rollout store) instead of being
    overwritten two steps later: the rows then have to drain to HBM, whereas two alternating
    buffers of the headline size stay in the 256 MiB last-level cache.  Reported next to the
    headline so that both regimes are on record; `step_many` is the multi-step launch
    (marl_env_step_many, 32 steps per launch) writing the same rows."""
    import torch

    dev = rew.device
    ring = int(max(4, min(ring, (8 << 30) // (N * D * 4))))  # at most 8 GB of rows
    store = torch.empty((ring, N, D), dtype=torch.float32, device=dev)
    R = actions.shape[0]
    for i in range(2 * ring):
        env.step_into(actions[i % R], store[i % ring], rew, term, trunc)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for i in range(steps):
        env.step_into(actions[i % R], store[i % ring], rew, term, trunc)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / steps
    out = {"obs_ring": ring, "us_per_launch": us, "value": N / us * 1e6, "frac": N * B / us / 1e3 / HBM_PEAK_GBS}
    if (N * D) % 4 == 0 and R >= ring:
        rew_k = torch.empty((ring, N), dtype=torch.float32, device=dev)
        te_k = torch.empty((ring, N), dtype=torch.bool, device=dev)
        tr_k = torch.empty_like(te_k)
        acts_k = actions[:ring].contiguous()
        reps = max(steps // ring, 2)
        env.step_many_into(acts_k, store, rew_k, te_k, tr_k)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            env.step_many_into(acts_k, store, rew_k, te_k, tr_k)
        ev1.record()
        torch.cuda.synchronize()
        us_m = ev0.elapsed_time(ev1) * 1e3 / (reps * ring)
        out["step_many"] = {"steps_per_launch": ring, "us_per_step": us_m, "value": N / us_m * 1e6,
                            "frac": N * B / us_m / 1e3 / HBM_PEAK_GBS}
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
    ap.add_argument("--graph", action="store_true", help="replay the steps from a hipGraph of 32 launches instead of launching eagerly")
    ap.add_argument("--no-graph", action="store_true", help="(default) launch every step eagerly from Python")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rows-kept", action="store_true",
                    help="skip the extra leg that times the kept-rows regime (profiling runs: keeps the traced launches uniform)")
    ap.add_argument("--obs-ring", type=int, default=2,
                    help="observation buffers the steps cycle through (2: the next step overwrites the rows the consumer "
                         "has just read, they never have to leave the last-level cache; 128: every step's rows are kept "
                         "like in a rollout store and drain to HBM)")
    ap.add_argument("--mode", choices=["random", "policy"], default="random",
                    help="random: pre-generated uniform actions (headline); policy: actions sampled from the "
                         "reference-shaped MLP (obs->512->256->7/1, fp32, PyTorch-ROCm) every step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    backend = os.environ.get("MARL_AMD_DIST_BACKEND", "nccl")  # gloo: several ranks on one GPU (rehearsal only)
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from marl_amd.env import BatchedOvercookedVecEnv
    from marl_amd.layouts import get_layout

    E, A = args.envs, args.agents
    rows = get_layout(args.layout)
    env = BatchedOvercookedVecEnv(rows, A, args.feature, E, max_steps=1000, seed=1234, env_index_base=rank * E,
                                  device=dev)
    N, D = env.N, env.D
    R = 32  # distinct pre-generated action batches, cycled
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    actions = torch.randint(0, 7, (R, N), dtype=torch.int64, device=dev, generator=gen)
    RING = max(2, args.obs_ring)
    obs = [torch.empty((N, D), dtype=torch.float32, device=dev) for _ in range(RING)]
    rew = torch.empty(N, dtype=torch.float32, device=dev)
    term = torch.empty(N, dtype=torch.bool, device=dev)
    trunc = torch.empty(N, dtype=torch.bool, device=dev)
    env.reset(out=obs[0])

    policy = None
    if args.mode == "policy":
        from marl_amd.policy import Agent

        torch.manual_seed(1)
        policy = Agent(env.single_observation_space, env.single_action_space, num_agents=A, num_envs=1,
                       build_centralised=False).to(dev)  # the centralised critic is not evaluated on this path
        policy.use_fused_sampling(1234, row_index_base=rank * N)  # sample + log_prob + entropy in one launch

    def eager(i):
        if policy is None:
            env.step_into(actions[i % R], obs[i % RING], rew, term, trunc)
        else:
            with torch.no_grad():
                act, _, _, _ = policy.get_action_and_value(obs[(i - 1) % RING])
            env.step_into(act, obs[i % RING], rew, term, trunc)

    for i in range(args.warmup):
        eager(i)
    torch.cuda.synchronize()

    graph = None
    if args.graph and not args.no_graph and args.steps >= R:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(R):
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
        for _ in range(args.steps // R):
            graph.replay()
        done = (args.steps // R) * R
    for i in range(done, args.steps):
        eager(i)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)

    if world > 1:
        tt = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, dev_ms = tt.tolist()

    if rank == 0:
        B = algorithmic_bytes_per_agent_step(D, env.state_words, A, actions.element_size())
        launch_s = dev_ms * 1e-3 / args.steps  # device-side time per launch, HIP events on the launch stream
        achieved = N * B / launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        key = f"{args.layout}/{A}/{args.feature}/{E}"
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
                                   + ", auto-reset at t=1000",
                       "launch": "hipGraph of 32 steps" if graph is not None else "eager"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_agent_step": B, "kernel": f"env_kernel<{A} agents, {args.feature}> (step+featurize)",
                         "us_per_launch": launch_s * 1e6},
        }
        out["config"]["obs_ring"] = RING
        if policy is not None:
            # the per-step time now contains the MLP forward pass: no single-kernel roofline applies
            out["roofline"] = None
        elif world == 1 and RING == 2 and not args.graph and not args.no_rows_kept:
            out["rows_kept"] = rows_kept_leg(env, actions, rew, term, trunc, N, D, B)
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(rows, A, args.feature, E, 1234, threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

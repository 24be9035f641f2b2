"""Training entry point with the reference's command line (main.py:99-171) for the MAPPO /
centralised-MAPPO path:

    python -m marl_amd.train --num-agents 2 --num-envs 16 --layout overcooked_cramped_room_v0 \
        --num-steps 256 --num-minibatches 4 --total-steps 20000000 --ppo-epoch 5 --clip-param 0.05 \
        --value-loss-coef 0.1 --entropy-coef 0.01 --lr 3e-4 --feature global_obs

Multi-GPU: ``python -m torch.distributed.run --nproc-per-node G -m marl_amd.train ...``; every rank
owns ``--num-envs`` envs (global env index = rank * num_envs + e) and an identical policy
replica; gradients are averaged over RCCL once per minibatch.  QMIX / SARSA / SAC are not part
of this build.
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import time

import numpy as np
import torch

from . import dist_utils
from .agents import CMAPPO, MAPPO
from .env import OBS_KEY, make_env, make_vector_env
from .policy import Agent
from .rollout import agent_environment_loop, make_writer
from .rollout_buffer import Buffer


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--cuda", action="store_true", default=False, help="accepted for compatibility; a GPU is required")
    p.add_argument("--num-agents", type=int, default=4)
    p.add_argument("--num-envs", type=int, default=8, help="envs per rank")
    p.add_argument("--layout", type=str, default="large_overcooked_layout")
    p.add_argument("--save-path", type=str, default=None)
    p.add_argument("--data-path", type=str, default="data")
    p.add_argument("--save", action="store_true", default=False)
    p.add_argument("--total-steps", type=int, default=1000, help="agent-steps per rank")
    p.add_argument("--num-steps", type=int, default=128)
    p.add_argument("--num-minibatches", type=int, default=4)
    p.add_argument("--log", action="store_true", default=False)
    p.add_argument("--render", action="store_true", default=False)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--feature", type=str, default="global_obs")
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--centralised", action="store_true", default=False)
    p.add_argument("--algorithm", type=str, default="mappo", choices=["mappo", "cmappo", "qmix", "sarsa", "sac"])
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--ppo-epoch", type=int, default=10)
    p.add_argument("--clip-param", type=float, default=0.2)
    p.add_argument("--value-loss-coef", type=float, default=0.5)
    p.add_argument("--entropy-coef", type=float, default=0.01)
    p.add_argument("--max-grad-norm", type=float, default=0.5)
    p.add_argument("--lam", type=float, default=0.95)
    p.add_argument("--max-steps", type=int, default=1000, help="episode length (overcooked_config.py:108)")
    p.add_argument("--shuffle", choices=["device", "numpy"], default="device")
    p.add_argument("--sampler", choices=["fused", "torch"], default="fused",
                   help="rollout action sampling: one HIP launch with a counter RNG, or torch.distributions")
    p.add_argument("--graph-update", choices=["auto", "on", "off"], default="auto",
                   help="replay the PPO minibatch step from a hipGraph (auto: single rank and minibatches of at most 65,536 samples)")
    p.add_argument("--adam", choices=["fused", "foreach"], default="fused",
                   help="Adam step as one multi-tensor launch (GPU only) or PyTorch's default per-operation launches")
    p.add_argument("--fused-loss", choices=["on", "off"], default="on",
                   help="decentralised MAPPO on the GPU: loss and head gradients from one HIP pass instead of autograd's small ops")
    p.add_argument("--env-shards", type=int, default=1,
                   help="run the envs of a rank as this many shards on HIP streams of their own; with the decentralised "
                        "critic and the fused sampler every shard then runs act -> step on its stream and one shard's "
                        "policy GEMMs overlap the other's env kernel (same stored rows as with 1)")
    p.add_argument("--graph-rollout", choices=["auto", "on", "off"], default="auto",
                   help="replay rollout steps from hipGraphs (auto: at most 16,384 rows per launch, on any number of ranks; "
                        "RCCL collectives between graph replays have run with one rank only so far: if a multi-GPU job "
                        "stalls, 'off' restores eager rollouts)")
    return p


def write_csv(path, values):
    with open(path, "w") as f:
        f.write("0\n")  # pandas' header for an unnamed single column, what plot.py reads back
        for v in values:
            f.write(f"{v}\n")


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.algorithm in ("qmix", "sarsa", "sac"):
        raise SystemExit(f"--algorithm {args.algorithm} is outside this build (MAPPO / CMAPPO rollout path only)")
    if args.render:
        raise SystemExit("--render is not available: there is no renderer in this build")
    rank, local_rank, world = dist_utils.init_from_env()
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    batch_size = args.num_envs * args.num_agents * args.num_steps  # main.py:174
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)  # main.py:177-179

    env = make_env(args.num_agents, layout=args.layout, feature=args.feature, render_mode=None,
                   max_steps=args.max_steps)
    vec_env = make_vector_env(args.num_envs, env, shards=args.env_shards, seed=args.seed,
                              env_index_base=rank * args.num_envs, device=device)
    assert vec_env.num_envs == args.num_envs * args.num_agents  # main.py:186
    obs_space, action_space = vec_env.single_observation_space, vec_env.single_action_space
    log_dir = f"logs/run__{int(time.time())}"
    single_obs_dim = env.observation_spaces[0][OBS_KEY].shape
    single_action_dim = env.action_spaces[0].n

    # model.py:28-34 always builds the centralised critic, obs_dim * num_agents * num_envs wide.  Kept for
    # the centralised learner and whenever it is small (checkpoints then carry the reference's
    # keys); a decentralised run with many envs leaves it out instead of allocating gigabytes.
    central_weights = obs_space.shape[0] * args.num_agents * args.num_envs * 512
    if args.algorithm == "cmappo" and central_weights > 2 ** 33:
        raise SystemExit(f"--algorithm cmappo with --num-envs {args.num_envs}: the reference's centralised critic is "
                         f"obs_dim*num_agents*num_envs = {central_weights // 512} inputs wide "
                         f"({central_weights * 4 / 2 ** 30:.0f} GiB of weights); use fewer envs per rank")
    net = Agent(obs_space, action_space, num_agents=args.num_agents, num_envs=args.num_envs,
                build_centralised=args.algorithm == "cmappo" or central_weights <= 2 ** 26).to(device)
    dist_utils.broadcast_parameters(net)
    if args.sampler == "fused":
        net.use_fused_sampling(args.seed, row_index_base=rank * args.num_envs * args.num_agents)
    # main.py:294.  fused: the whole Adam step is one multi-tensor launch instead of seventeen
    # (same arithmetic; a minibatch step at the reference's batch sizes is launch-bound)
    optimizer = torch.optim.Adam(net.parameters(), lr=args.lr, betas=(0.9, 0.95),
                                 fused=device.type == "cuda" and args.adam == "fused")
    buffer = Buffer(single_obs_dim[0], env.config["num_agents"], args.num_envs, max_size=args.num_steps, device=device)
    # main.py:291,303: the centralised learner is reached only through --algorithm cmappo
    cls = CMAPPO if args.algorithm == "cmappo" else MAPPO
    writer = make_writer(log_dir) if (args.log and rank == 0) else None
    agent = cls(vec_env, optimizer, net, buffer, single_obs_dim, single_action_dim, batch_size=batch_size,
                num_mini_batches=args.num_minibatches, ppo_epoch=args.ppo_epoch, clip_param=args.clip_param,
                value_loss_coef=args.value_loss_coef, entropy_coef=args.entropy_coef,
                max_grad_norm=args.max_grad_norm, gamma=args.gamma, lam=args.lam, save_path=args.save_path,
                log_dir=log_dir, num_agents=args.num_agents, log=args.log and rank == 0, args=args,
                shuffle=args.shuffle, writer=writer, graph_update=args.graph_update,
                fused_loss=args.fused_loss == "on")
    num_updates = args.total_steps // batch_size  # main.py:302

    t0 = time.time()
    returns, freq = agent_environment_loop(agent, vec_env, device, num_update=num_updates, log_dir=log_dir, args=args,
                                           writer=writer, progress=(rank == 0 and sys.stderr.isatty()))
    torch.cuda.synchronize()
    dt = time.time() - t0
    if writer is not None:
        writer.close()  # tensorboard: flush the event file; without tensorboard: <log_dir>/scalars.csv
    if rank == 0:
        steps = num_updates * batch_size * world
        print(f"{steps} agent-steps in {dt:.1f} s ({steps / max(dt, 1e-9):.3e}/s), {len(returns)} episodes; "
              f"last returns (first env): {returns[-5:]}; all-env mean: {freq['episode_return_all_envs'][-5:]}")
        alg = "centralised" if args.centralised else "decentralised"  # main.py:328-338
        folder = args.data_path
        os.makedirs(folder, exist_ok=True)
        stem = f"{alg}_{args.num_agents}_{args.layout}"
        write_csv(os.path.join(folder, f"{stem}_returns_seed_{args.seed}.csv"), returns)
        for key in ("frequency_delivery_per_episode", "frequency_plated_per_episode",
                    "frequency_ingredient_in_pot_per_episode"):
            write_csv(os.path.join(folder, f"{stem}_{key}_seed_{args.seed}.csv"), freq[key])
        # not "..._returns_...": plot.py:112 reads every file of the folder whose name contains its keyword
        write_csv(os.path.join(folder, f"{stem}_all_envs_episode_return_seed_{args.seed}.csv"), freq["episode_return_all_envs"])
        with open(os.path.join(folder, f"{stem}_args_seed_{args.seed}.txt"), "w") as f:
            for k, v in vars(args).items():
                f.write(f"{k}: {v}\n")
            from .policy import gemm_choices

            # which library call ran each small policy layer (a fixed function of its shape unless overridden)
            f.write(f"small_gemm_forms: {sorted(gemm_choices().items())}\n")
        if args.save and args.save_path:
            print("saved", agent.save_model())
    return returns, freq


if __name__ == "__main__":
    main()

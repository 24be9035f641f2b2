"""Plays a saved policy, counterpart of the reference's tests/test_load.py:47-134
(`python3 test_load.py --model-path <policy.pth>`): loads a ``model.Agent`` state_dict (the
reference's key names; written by ``marl_amd.train --save`` or by the reference itself), runs
whole episodes and reports what the agents achieved.  Many envs are played at once on the GPU
instead of one rendered env; there is no renderer in this build.

    python -m marl_amd.evaluate --model-path models/decentralised_policy_2_agents_overcooked_cramped_room_v0_seed_1.pth \
        --num-agents 2 --layout overcooked_cramped_room_v0 --feature global_obs
"""
from __future__ import annotations

import argparse
import json

import torch

from .env import OBS_KEY, make_env, make_vector_env
from .episode_stats import EpisodeStats
from .policy import Agent


def load_policy(path, obs_space, action_space, num_agents, device):
    """state_dict -> Agent.  ``weights_only=True``: nothing in the file is executed.  The
    centralised critic is not needed for acting; its weights are ignored when present."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = {k: v for k, v in sd.items() if not k.startswith("centralised_critics")}
    net = Agent(obs_space, action_space, num_agents=num_agents, num_envs=1, build_centralised=False)
    net.load_state_dict(sd)
    return net.to(device).eval()


def run_inference(net, vec, episodes=1, greedy=False):
    """tests/test_load.py:113-134 for every env at once; returns per-episode averages over envs."""
    E, A = vec.E, vec.A
    stats = EpisodeStats(E, A, vec.device)
    out = []
    for _ in range(episodes):
        obs = vec.reset()[0][OBS_KEY]
        stats.reset()
        for _ in range(vec.max_steps):
            with torch.no_grad():
                dist = net.get_prob(obs)
                actions = dist.logits.argmax(1) if greedy else dist.sample()
            o, rewards, _, _, _ = vec.step(actions)
            obs = o[OBS_KEY]
            stats.add(rewards)
        d = stats.read()
        out.append({"return": d["all_return"], "deliveries": d["all_deliveries"], "platings": d["all_platings"],
                    "onions_in_pot": d["all_pottings"]})
    return out


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--model-path", type=str, required=True)
    p.add_argument("--layout", type=str, default="overcooked_cramped_room_v0")
    p.add_argument("--num-agents", type=int, default=2)
    p.add_argument("--feature", type=str, default="global_obs")
    p.add_argument("--num-envs", type=int, default=256)
    p.add_argument("--episodes", type=int, default=1)
    p.add_argument("--max-steps", type=int, default=1000)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--greedy", action="store_true", help="argmax actions instead of sampling (the reference samples)")
    args = p.parse_args(argv)
    device = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(args.seed)
    env = make_env(args.num_agents, layout=args.layout, feature=args.feature, max_steps=args.max_steps)
    vec = make_vector_env(args.num_envs, env, seed=args.seed, device=device)
    net = load_policy(args.model_path, vec.single_observation_space, vec.single_action_space, args.num_agents, device)
    result = run_inference(net, vec, episodes=args.episodes, greedy=args.greedy)
    for k, r in enumerate(result):
        print(json.dumps({"episode": k, "envs": args.num_envs, **r}))
    return result


if __name__ == "__main__":
    main()

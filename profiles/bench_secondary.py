"""Secondary measurements (not the headline bench): GAE kernel GB/s, the other BASELINE.json
configs' step throughput, reset / featurize-only launches.  Prints one JSON object."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd.env import BatchedOvercookedVecEnv  # noqa: E402
from marl_amd.gae import compute_gae  # noqa: E402
from marl_amd.layouts import get_layout  # noqa: E402


def timed(fn, iters, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def step_rate(layout, A, feature, E, iters=512):
    env = BatchedOvercookedVecEnv(get_layout(layout), A, feature, E, seed=1)
    N, D = env.N, env.D
    obs = [torch.empty((N, D), device="cuda") for _ in range(2)]
    rew = torch.empty(N, device="cuda")
    term = torch.empty(N, dtype=torch.bool, device="cuda")
    trunc = torch.empty_like(term)
    acts = torch.randint(0, 7, (16, N), device="cuda")
    env.reset(out=obs[0])
    k = [0]

    def fn():
        env.step_into(acts[k[0] % 16], obs[k[0] % 2], rew, term, trunc)
        k[0] += 1

    dt = timed(fn, iters, warm=32)
    B = 4 * D + 6 + 8 + 2 * 4 * env.state_words / A
    t_reset = timed(lambda: env.reset(out=obs[0]), 64)
    t_feat = timed(lambda: env.featurize(out=obs[1]), 64)
    return {"layout": layout, "agents": A, "feature": feature, "envs": E, "D": D, "state_words": env.state_words,
            "us_per_step": dt * 1e6, "agent_steps_per_s": N / dt, "bytes_per_agent_step": B,
            "achieved_GBps": N * B / dt / 1e9, "us_reset": t_reset * 1e6, "us_featurize": t_feat * 1e6}


def gae_rate(T, N, mode):
    r = torch.rand(T, N, device="cuda")
    d = (torch.rand(T, N, device="cuda") < 0.01).float()
    v = torch.randn(T, N, device="cuda")
    nv = torch.randn(1, 1 if mode else N, device="cuda")
    dt = timed(lambda: compute_gae(r, d, v, nv, 0.99, 0.95, centralised=bool(mode)), 50)
    bytes_ = T * N * 20.0
    return {"T": T, "N": N, "mode": "cmappo" if mode else "mappo", "us": dt * 1e6, "achieved_GBps": bytes_ / dt / 1e9,
            "frac_of_8TBps": bytes_ / dt / 8e12}


if __name__ == "__main__":
    out = {"step": [
        step_rate("overcooked_cramped_room_v0", 2, "global_obs", 65536),
        step_rate("overcooked_forced_coordination_v0", 2, "global_obs", 65536),
        step_rate("overcooked_counter_circuit_v0", 2, "global_obs", 131072),
        step_rate("overcooked_coordination_ring_v0", 4, "local_obs", 65536),
        step_rate("overcooked_coordination_ring_v0", 4, "global_obs", 32768),
        step_rate("large_overcooked_layout", 4, "global_obs", 32768),
        step_rate("overcooked_cramped_room_v0", 2, "local_obs", 65536),
        step_rate("overcooked_cramped_room_v0", 2, "Minimal_spatial", 65536),
        step_rate("overcooked_cramped_room_v0", 2, "global_obs", 1048576, iters=64),
    ], "gae": [gae_rate(128, 131072, 0), gae_rate(128, 131072, 1), gae_rate(256, 32, 0)]}
    print(json.dumps(out, indent=1))

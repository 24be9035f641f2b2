"""Long bit-for-bit comparison of the HIP env with the CPU oracle at large env counts (one-off
evidence run, too long for the test suite): every step compares all observation rows, rewards and
flags, every 50th step and every restart the exported state.

    python profiles/diag/soak_parity.py [steps] [first case] [last case + 1]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402  (the checker)
from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv  # noqa: E402
from marl_amd.layouts import get_layout  # noqa: E402

CASES = [
    # layout, agents, feature, envs, max_steps
    ("overcooked_cramped_room_v0", 2, "global_obs", 65536, 400),
    ("overcooked_counter_circuit_v0", 2, "global_obs", 32768, 300),
    ("overcooked_forced_coordination_v0", 2, "local_obs", 32768, 250),
    ("overcooked_coordination_ring_v0", 4, "local_obs", 16384, 300),
    ("large_overcooked_layout", 4, "global_obs", 8192, 350),
    ("overcooked_coordination_ring_v0", 3, "global_obs", 16384, 200),
    # second batch: one agent, the remaining feature sets and layouts
    ("overcooked_cramped_room_v0", 1, "global_obs", 32768, 150),
    ("overcooked_asymmetric_advantages_v0", 2, "Minimal_spatial", 32768, 250),
    ("overcooked_counter_circuit_v0", 2, "Minimal_spatial_other_agent_aware", 32768, 300),
    ("overcooked_cramped_room_v0", 4, "only_direction", 16384, 120),
    ("overcooked_asymmetric_advantages_v0", 3, "local_obs", 16384, 200),
    ("overcooked_forced_coordination_v0", 2, "global_obs", 65536, 333),
]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    oracle.build()
    threads = min(16, len(os.sched_getaffinity(0)))
    dev = torch.device("cuda", 0)
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(CASES)
    for layout, A, feature, E, max_steps in CASES[lo:hi]:
        rows = get_layout(layout)
        gpu = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=max_steps, seed=99, env_index_base=12345, device=dev)
        ora = oracle.OracleVecEnv(rows, A, feature, max_steps=max_steps, num_envs=E, seed=99, env_index_base=12345,
                                  nthreads=threads)
        og, _ = gpu.reset()
        oo = ora.reset()
        assert torch.equal(og[OBS_KEY], torch.from_numpy(oo).to(dev))
        rng = np.random.default_rng(5)
        t0 = time.perf_counter()
        reward_sum, restarts = 0.0, 0
        for t in range(steps):
            if t % 3 == 0:
                act = rng.integers(0, 7, size=E * A).astype(np.int64)
            else:  # lean on interact so that soups get cooked and delivered
                act = rng.choice(7, size=E * A, p=[0.14, 0.14, 0.14, 0.14, 0.3, 0.04, 0.1]).astype(np.int64)
            o_g, r_g, te_g, tr_g, _ = gpu.step(torch.from_numpy(act).to(dev))
            o_o, r_o, te_o, tr_o = ora.step(act)
            if not torch.equal(o_g[OBS_KEY], torch.from_numpy(o_o).to(dev)):
                raise SystemExit(f"{layout}/{A}/{feature}: observations differ at step {t}")
            if not (np.array_equal(r_g.cpu().numpy(), r_o) and np.array_equal(te_g.cpu().numpy(), te_o)
                    and np.array_equal(tr_g.cpu().numpy(), tr_o)):
                raise SystemExit(f"{layout}/{A}/{feature}: rewards or flags differ at step {t}")
            reward_sum += float(r_o.sum())
            if tr_o.any():
                restarts += 1
            if t % 50 == 0 or tr_o.any():
                snap = gpu.export_state()
                for k, ref in (("agents", ora.agents), ("pots", ora.pots), ("items", ora.items), ("misc", ora.misc)):
                    if not np.array_equal(snap[k], ref):
                        raise SystemExit(f"{layout}/{A}/{feature}: state '{k}' differs at step {t}")
        dt = time.perf_counter() - t0
        print(f"{layout} {A} agents {feature} {E} envs: {steps} steps, {E * A * steps:.3g} agent-steps, "
              f"{E * A * steps * gpu.D:.3g} observation floats compared, {restarts} batch restarts, "
              f"reward sum {reward_sum:.1f}: identical ({dt:.0f} s)", flush=True)
        del gpu, ora


if __name__ == "__main__":
    main()

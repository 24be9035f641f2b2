"""Random shapes through the GAE kernel and the reward-event passes against the CPU restatements
(one-off evidence run beyond the suite's fixed shapes): decentralised GAE bit for bit, centralised GAE
to 1e-5 of the tensor's scale, reward events exact.

    python profiles/diag/fuzz_small_passes.py [cases]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402  (the checker)
from marl_amd.episode_stats import EpisodeStats  # noqa: E402
from marl_amd.gae import compute_gae  # noqa: E402

POSSIBLE = np.array([float(np.float32(own + k)) for own in (0.0, 0.1, 0.3) for k in range(0, 5)], dtype=np.float32)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    oracle.build()
    dev = "cuda:0"
    t0 = time.perf_counter()
    samples = 0
    for case in range(cases):
        rng = np.random.default_rng(70_000 + case + int(os.environ.get("FUZZ_BASE", 0)))
        T = int(rng.choice([1, 2, 3, 5, 8, 16, 33, 128, 257, 400]))
        N = int(rng.choice([1, 2, 3, 31, 64, 65, 255, 1000, 4097, 32768, 70001]))
        if T * N > 6_000_000:
            N = max(1, 6_000_000 // T)
        mode = int(rng.integers(0, 2))
        gamma, lam = float(rng.choice([0.0, 0.9, 0.99, 1.0])), float(rng.choice([0.0, 0.5, 0.95, 1.0]))
        rewards = np.where(rng.random((T, N)) < rng.choice([0.0, 0.1, 1.0]), rng.random((T, N)) * 2, 0).astype(np.float32)
        dones = (rng.random((T, N)) < rng.choice([0.0, 0.02, 0.5, 1.0])).astype(np.float32)
        values = (rng.standard_normal((T, N)) * rng.choice([0.01, 1.0, 50.0])).astype(np.float32)
        nv = rng.standard_normal(1 if mode else N).astype(np.float32)
        adv_o, ret_o = oracle.gae(rewards, dones, values, nv, gamma, lam, mode)
        adv, ret = compute_gae(torch.from_numpy(rewards).to(dev), torch.from_numpy(dones).to(dev), torch.from_numpy(values).to(dev),
                               torch.from_numpy(nv).to(dev).reshape(1, -1), gamma, lam, centralised=bool(mode))
        adv, ret = adv.cpu().numpy(), ret.cpu().numpy()
        tag = f"case {case}: T={T} N={N} mode={mode} gamma={gamma} lam={lam}"
        if mode == 0:
            if not (np.array_equal(adv, adv_o) and np.array_equal(ret, ret_o)):
                print("GAE differs:", tag)
                return 1
        else:
            scale = max(float(np.abs(adv_o).max()), 1e-30)
            if not (np.allclose(adv, adv_o, rtol=1e-5, atol=1e-5 * scale) and np.allclose(ret, ret_o, rtol=1e-5, atol=1e-5 * scale)):
                print("centralised GAE differs:", tag, float(np.abs(adv - adv_o).max()), scale)
                return 1
        samples += T * N
        # reward events: K stored steps at once and step by step
        E, A = int(rng.choice([1, 2, 7, 256, 257, 5000, 40000])), int(rng.integers(1, 5))
        K = int(rng.choice([1, 2, 9, 40]))
        p = np.full(POSSIBLE.size, 0.03)
        p[0] = 1.0 - 0.03 * (POSSIBLE.size - 1)
        r = rng.choice(POSSIBLE, size=(K, E * A), p=p).astype(np.float32)
        totals, first = np.zeros(4, np.int64), np.zeros(4, np.float64)
        for k in range(K):
            t, f = oracle.reward_events(r[k], E, A)
            totals += t
            first += f
        for form in ("many", "single"):
            stats = EpisodeStats(E, A, dev)
            rd = torch.from_numpy(r).to(dev)
            if form == "many":
                stats.add_many(rd)
            else:
                for k in range(K):
                    stats.add(rd[k])
            got_t, got_f = stats.raw()
            if not (np.array_equal(got_t.cpu().numpy(), totals) and np.array_equal(got_f.cpu().numpy(), first)):
                print(f"reward events ({form}) differ: case {case}: E={E} A={A} K={K}")
                return 1
        if case % 50 == 49:
            print(f"{case + 1} cases, {samples:.3g} GAE samples identical / within tolerance, {time.perf_counter() - t0:.0f} s", flush=True)
    print(f"{cases} random cases: decentralised GAE bit for bit, centralised GAE within 1e-5 of the scale, reward events (single "
          f"steps and many at once) exact; {samples:.3g} GAE samples ({time.perf_counter() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Differential fuzz of the HIP env against the CPU oracle over random kitchens (one-off evidence run,
longer and wider than the suite's 24 cases): grids up to 14 x 14, 1-4 agents, every feature set, both
distance metrics, drawn and caller-given serving orders, short episodes so that envs restart many times.
Both sides must also agree on which layouts they refuse.  Every 25 steps, in half of the cases, a random valid
state is injected into both (items on any share of the cells, pots in every condition, any holdings): its
observations are compared at once and play goes on from it.

    python profiles/diag/fuzz_kitchens.py [cases] [first seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402  (the checker)
from marl_amd import _lib  # noqa: E402
from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv  # noqa: E402

FEATURES = ["global_obs", "local_obs", "Minimal_spatial", "Minimal_spatial_other_agent_aware", "only_direction"]


def random_layout(rng):
    H, W = int(rng.integers(3, 15)), int(rng.integers(3, 15))
    density = rng.choice([0.35, 0.6, 0.8, 0.95])  # share of floor among the inner cells
    rest = (1.0 - density) / 6.0
    grid = [["#"] * W for _ in range(H)]
    pots = 0
    spawn_marks = rng.random() < 0.25
    for r in range(H):
        for c in range(W):
            border = r in (0, H - 1) or c in (0, W - 1)
            ch = rng.choice(list(" CUO=@#"), p=[0.0, 0.3, 0.2, 0.15, 0.15, 0.15, 0.05] if border
                            else [density, rest, rest, rest, rest, rest, rest])
            if ch == "U":
                if pots == 4:
                    ch = "C"
                else:
                    pots += 1
            if ch == " " and spawn_marks and rng.random() < 0.1:
                ch = "+"
            grid[r][c] = str(ch)
    return ["".join(row) for row in grid]


def random_snapshot(rng, rows, A, E, NP, max_steps):
    """a valid state that random play may never reach: items on any share of the floor and counter cells,
    pots in every condition, any holdings"""
    H, W = len(rows), len(rows[0])
    flat = "".join(rows)
    floor = np.array([k for k, ch in enumerate(flat) if ch in " +"])
    holder = np.array([k for k, ch in enumerate(flat) if ch in " +C"])
    agents = np.zeros((E, A, 4), np.int32)
    items = np.zeros((E, H * W), np.uint8)
    pots = np.zeros((E, NP, 2), np.int32)
    misc = np.zeros((E, 2), np.int32)
    fill = rng.choice([0.0, 0.1, 0.5, 1.0])
    for e in range(E):
        cells = rng.choice(floor, size=A, replace=False)
        agents[e, :, 0], agents[e, :, 1] = cells // W, cells % W
        agents[e, :, 2] = rng.integers(0, 4, size=A)
        agents[e, :, 3] = rng.integers(0, 4, size=A)
        put = holder[rng.random(holder.size) < fill]
        items[e, put] = rng.integers(1, 4, size=put.size)
        items[e, cells] = 0
        for p in range(NP):
            c = int(rng.integers(0, 4))
            pots[e, p] = (c, int(rng.integers(0, 31)) if c == 3 else 30)
        misc[e] = (int(rng.integers(0, max_steps)), int(rng.integers(0, 1 << 20)))
    return {"agents": agents, "pots": pots, "items": items, "misc": misc}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle.build()
    dev = torch.device("cuda", 0)
    ran = refused = capacity = injected = 0
    agent_steps = 0
    t0 = time.perf_counter()
    for case in range(first, first + cases):
        rng = np.random.default_rng(50_000 + case)
        rows = random_layout(rng)
        A = int(rng.integers(1, 5))
        feature = FEATURES[int(rng.integers(0, len(FEATURES)))]
        metric = int(rng.integers(0, 2))
        E = int(rng.choice([1, 3, 17, 64, 130]))
        max_steps = int(rng.integers(5, 60))
        steps = 100
        use_prio = rng.random() < 0.3
        gpu = ora = None
        try:
            gpu = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=max_steps, seed=case, env_index_base=case * 7,
                                          dist_metric=metric, device=dev)
        except _lib.MarlAmdError as ex:
            gpu_err = str(ex)
        try:
            ora = oracle.OracleVecEnv(rows, A, feature, max_steps=max_steps, num_envs=E, seed=case, env_index_base=case * 7,
                                      dist_metric=metric)
        except ValueError as ex:
            ora_err = str(ex)
        if gpu is None and ora is not None and "more than" in gpu_err:
            capacity += 1  # a documented capacity limit of the kernel's tables (item-capable cells, floor cells), not a rule
            continue
        if gpu is None or ora is None:
            if (gpu is None) != (ora is None):
                print(f"case {case}: only one side refuses {rows} with {A} agents: "
                      f"gpu {'refused: ' + gpu_err if gpu is None else 'accepted'}, oracle {'refused: ' + ora_err if ora is None else 'accepted'}")
                return 1
            refused += 1
            continue
        tag = f"case {case} ({len(rows)}x{len(rows[0])}, {A} agents, {feature}, metric {metric}, {E} envs, max_steps {max_steps}, prio {use_prio}) {rows}"
        og = gpu.reset()[0][OBS_KEY].cpu().numpy()
        oo = ora.reset()
        if not np.array_equal(og, oo):
            print("reset observations differ:", tag)
            return 1
        for t in range(steps):
            act = rng.choice(7, size=E * A, p=[0.14, 0.14, 0.14, 0.14, 0.3, 0.04, 0.1]).astype(np.int64)
            prio = rng.integers(0, 4, size=(E, A)).astype(np.uint8) if use_prio else None
            o_g, r_g, te_g, tr_g, _ = gpu.step(torch.from_numpy(act), prio_override=None if prio is None else torch.from_numpy(prio))
            o_o, r_o, te_o, tr_o = ora.step(act, prio_override=prio)
            if not (np.array_equal(o_g[OBS_KEY].cpu().numpy(), o_o) and np.array_equal(r_g.cpu().numpy(), r_o)
                    and np.array_equal(tr_g.cpu().numpy(), tr_o) and np.array_equal(te_g.cpu().numpy(), te_o)):
                print(f"step {t} differs:", tag)
                return 1
            if t % 25 == 24:
                snap = gpu.export_state()
                if not (np.array_equal(snap["agents"], ora.agents) and np.array_equal(snap["pots"], ora.pots)
                        and np.array_equal(snap["items"], ora.items) and np.array_equal(snap["misc"], ora.misc)):
                    print(f"state after step {t} differs:", tag)
                    return 1
                if rng.random() < 0.5:
                    snap = random_snapshot(rng, rows, A, E, gpu.NP, max_steps)
                    gpu.import_state(snap)
                    ora.load_snapshot(snap)
                    injected += 1
                    if not np.array_equal(gpu.featurize().cpu().numpy(), ora.featurize()):
                        print(f"observations of the state injected after step {t} differ:", tag)
                        return 1
        agent_steps += E * A * steps
        ran += 1
        gpu.close()
        if (case - first) % 50 == 49:
            print(f"{case - first + 1} cases: {ran} run, {refused} refused by both sides, {agent_steps:.3g} agent-steps identical, "
                  f"{time.perf_counter() - t0:.0f} s", flush=True)
    print(f"seeds {first}..{first + cases - 1}: {ran} kitchens run, {refused} refused by both sides, {capacity} beyond the library's table sizes, {injected} random states injected, {agent_steps:.3g} agent-steps, "
          f"every observation, reward and flag of every step and the state every 25 steps: identical ({time.perf_counter() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())

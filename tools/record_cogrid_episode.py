#!/usr/bin/env python
"""Records episodes of the reference's real env (cogrid's Overcooked) in the format that
``python -m marl_amd.replay`` replays through the HIP env and compares array by array.

CANNOT RUN IN THE BUILD CONTAINER: cogrid is not installed there.  Run it where the reference runs:

    python tools/record_cogrid_episode.py --reference /path/to/MARL --layout overcooked_cramped_room_v0 \
        --num-agents 2 --feature global_obs --envs 8 --steps 400 --out cramped.npz
    python -m marl_amd.replay cramped.npz          # on the GPU box

The env is built the way the reference builds it (main.py:47-64); only attributes the reference's own
code reads are touched: ``env.grid.grid_agents[i].pos / .dir / .inventory``, ``env.grid.grid`` (objects
with ``.pos`` and ``.obj_placed_on``), ``env.grid.get(row, col)`` (overcooked_features.py:455-488), the
per-agent observation / reward / truncation dicts (tests/test_load.py:74-128).  Pot contents are read
through ``objects_in_pot`` / ``cooking_timer`` when the pot object has them (else ``pots`` is left out).
What cannot be seen from outside is the order in which the env served agents that moved in one
step.  ``serve_rank`` is therefore reconstructed from the outcome: a mover that ended on a new cell
gets rank 0, a mover that stayed rank 1 (whoever lost a contested cell was served after the winner).
``contested[k, e]`` marks the steps where some mover stayed put, so that a difference on such a step
can be told from a difference in the rules.
"""
import argparse
import sys

import numpy as np

ITEM_CODE = {"Onion": 1, "Plate": 2, "OnionSoup": 3}
CELL_CHAR = {"Wall": "#", "Counter": "C", "Pot": "U", "OnionStack": "O", "PlateStack": "=", "DeliveryZone": "@"}
MOVES = (0, 1, 2, 3)  # MoveUp, MoveDown, MoveLeft, MoveRight of the cardinal action set


def _name(obj):
    return type(obj).__name__ if obj is not None else None


def _shape(grid):
    if hasattr(grid, "height") and hasattr(grid, "width"):
        return int(grid.height), int(grid.width)
    pos = np.array([o.pos for o in grid.grid if o is not None])
    return int(pos[:, 0].max()) + 1, int(pos[:, 1].max()) + 1


def snapshot(env, agent_ids):
    """(agents [A,4], pots [NP,2] or None, items [H*W], layout rows) of one env"""
    grid = env.grid
    H, W = _shape(grid)
    agents = np.zeros((len(agent_ids), 4), np.int32)
    for k, aid in enumerate(agent_ids):
        a = grid.grid_agents[aid]
        held = [ITEM_CODE.get(_name(o), 0) for o in a.inventory]
        agents[k] = (int(a.pos[0]), int(a.pos[1]), int(a.dir), held[0] if held else 0)
    items = np.zeros(H * W, np.int32)
    rows, pots = [], []
    for r in range(H):
        row = ""
        for c in range(W):
            obj = grid.get(r, c)
            name = _name(obj)
            row += CELL_CHAR.get(name, " ")
            if name in ITEM_CODE:                       # lying loose on the floor
                items[r * W + c] = ITEM_CODE[name]
            elif name == "Counter":
                items[r * W + c] = ITEM_CODE.get(_name(getattr(obj, "obj_placed_on", None)), 0)
            elif name == "Pot":
                if hasattr(obj, "objects_in_pot") and hasattr(obj, "cooking_timer"):
                    pots.append((len(obj.objects_in_pot), int(obj.cooking_timer)))
                else:
                    pots = None
        rows.append(row)
    return agents, (np.array(pots, np.int32).reshape(-1, 2) if pots is not None else None), items, rows


def record_episode(env, steps, rng, obs_key="n_agent_overcooked_features"):
    """one env, one episode of at most `steps` uniformly random actions; dict of per-step arrays"""
    obs, _ = env.reset()
    ids = list(env.agent_ids)
    A = len(ids)
    agents, pots, items, rows = snapshot(env, ids)
    out = {"layout": rows, "spawn_cells": agents[:, :2].copy(), "spawn_dirs": agents[:, 2].copy(),
           "obs": [np.stack([obs[i][obs_key] for i in ids])], "agents": [agents], "items": [items],
           "pots": None if pots is None else [pots], "actions": [], "rewards": [], "truncated": [], "serve_rank": [],
           "contested": []}
    for _ in range(steps):
        act = rng.integers(0, 7, size=A)
        before = out["agents"][-1][:, :2]
        obs, rew, term, trunc, _ = env.step({aid: int(act[k]) for k, aid in enumerate(ids)})
        agents, pots, items, _ = snapshot(env, ids)
        moved = (agents[:, :2] != before).any(axis=1)
        wanted = np.isin(act, MOVES)
        out["serve_rank"].append(np.where(wanted & ~moved, 1, 0))
        out["contested"].append(bool((wanted & ~moved).any()))
        out["actions"].append(act)
        out["obs"].append(np.stack([obs[i][obs_key] for i in ids]))
        out["rewards"].append(np.array([rew[i] for i in ids], np.float32))
        out["truncated"].append(np.array([bool(trunc[i]) for i in ids]))
        out["agents"].append(agents)
        out["items"].append(items)
        if out["pots"] is not None:
            out["pots"] = None if pots is None else out["pots"] + [pots]
        if all(term[i] or trunc[i] for i in ids):
            break
    return out


def record(make_env, num_envs, steps, seed, feature, max_steps):
    """`num_envs` episodes from fresh envs, stacked into the replay format ([K(+1), E, A, ...])"""
    rng = np.random.default_rng(seed)
    eps = [record_episode(make_env(), steps, rng) for _ in range(num_envs)]
    K = min(len(e["actions"]) for e in eps)  # episodes are cut to the shortest
    rec = {"layout": np.array(eps[0]["layout"]), "feature": feature, "max_steps": max_steps,
           "spawn_cells": np.stack([e["spawn_cells"] for e in eps]), "spawn_dirs": np.stack([e["spawn_dirs"] for e in eps])}
    for name, extra in (("actions", 0), ("serve_rank", 0), ("rewards", 0), ("truncated", 0), ("contested", 0),
                        ("obs", 1), ("agents", 1), ("items", 1), ("pots", 1)):
        if all(e[name] is not None for e in eps):
            rec[name] = np.stack([np.stack(e[name][:K + extra]) for e in eps], axis=1)
    return rec


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="checkout of the reference (its overcooked_config.py registers features and layouts)")
    ap.add_argument("--layout", default="overcooked_cramped_room_v0")
    ap.add_argument("--num-agents", type=int, default=2)
    ap.add_argument("--feature", default="global_obs")
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    import functools

    from cogrid.envs import registry                   # main.py:4-7
    from cogrid.envs.overcooked import overcooked
    from overcooked_config import N_agent_overcooked_config

    config = N_agent_overcooked_config.copy()           # main.py:47-50
    config["num_agents"], config["features"] = args.num_agents, args.feature
    config["grid"] = dict(config["grid"], layout=args.layout)
    registry.register("NAgentOvercooked-V0", functools.partial(overcooked.Overcooked, config=config))  # main.py:55-60
    rec = record(lambda: registry.make("NAgentOvercooked-V0", render_mode=None), args.envs, args.steps, args.seed,
                 args.feature, int(config["max_steps"]))
    np.savez_compressed(args.out, **rec)
    print(f"wrote {args.out}: {rec['actions'].shape[0]} steps x {args.envs} envs x {args.num_agents} agents"
          + ("" if "pots" in rec else " (pot contents not readable: 'pots' left out)"))


if __name__ == "__main__":
    main()

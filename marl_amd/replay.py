"""Replays a recorded episode through the HIP env and compares it with the recording: the
harness for pinning the dynamics against the reference's third-party env (cogrid), whose
random choices (spawn cells, initial directions, the order in which moving agents are served)
are fed back as explicit inputs so that no random draw is made here.

A recording is an ``.npz`` (or a dict) of one episode over E envs and K steps:

  layout        [H] str       the ASCII grid (or ``layout_name``: a registered name)
  feature       str           feature set name (``global_obs`` ...)
  max_steps     int           truncation horizon of the env (K must not exceed it)
  spawn_cells   int [E, A, 2] (row, col) of every agent after reset
  spawn_dirs    int [E, A]    initial direction, 0 right 1 down 2 left 3 up
  actions       int [K, E, A] what each agent did (0 up 1 down 2 left 3 right 4 pickup/drop 5 toggle 6 noop)
  serve_rank    int [K, E, A] optional: rank in which the env served the agents that moved
                              (lower first; agents that did not move: any value)
  obs           f32 [K+1, E, A, D]  optional: observation after reset and after every step
  rewards       f32 [K, E, A]       optional
  truncated     bool [K, E, A]      optional
  agents        int [K+1, E, A, 4]  optional: (row, col, dir, holding) with holding 0 none 1 onion 2 plate 3 soup
  pots          int [K+1, E, NP, 2] optional: (onions in the pot, cooking timer)
  items         int [K+1, E, H*W]   optional: what lies on each cell, same codes as holding
  contested     bool [K, E]         optional: steps on which some agent that wanted to move did not (its
                                    serving order may have been reconstructed from the outcome)

``tools/record_cogrid_episode.py`` writes this format from the reference's real env.

Whatever optional array is present is compared, exactly.  ``python -m marl_amd.replay file.npz``
prints the first difference per array (step, env, agent and, for observations, the feature the
differing position belongs to) and exits non-zero if there is one.  All K steps run as one
launch (`marl_env_step_many`).
"""
from __future__ import annotations

import sys

import numpy as np
import torch

from .env import BatchedOvercookedVecEnv
from .layouts import get_layout

# own-block layout of the full feature sets (overcooked_features.py:38-78 for A agents): name, width
_FULL_BLOCK = (("agent_dir", 4), ("inventory", 3), ("next_to_counter", 4), ("next_to_pot", 16),
               ("closest_onion", 8), ("closest_plate", 8), ("closest_plate_stack", 4), ("closest_onion_stack", 4),
               ("closest_soup", 8), ("closest_delivery", 4), ("closest_counter", 8), ("pot_0", 11), ("pot_1", 11))


def feature_at(feature: str, num_agents: int, index: int) -> str:
    """name of the feature an observation position belongs to (full feature sets; otherwise the index)"""
    if feature not in ("global_obs", "local_obs"):
        return f"position {index}"
    tail = (("dist_to_other_agents", 2 * (num_agents - 1)), ("agent_position", 2), ("can_move_direction", 4))
    blk = sum(w for _, w in _FULL_BLOCK + tail)
    who, off = divmod(index, blk)
    for name, w in _FULL_BLOCK + tail:
        if off < w:
            owner = "own block" if who == 0 else f"block of other agent #{who}"
            return f"{name}[{off}] in the {owner}"
        off -= w
    return f"position {index}"


def load_recording(rec):
    """A recording (path of an .npz, or a dict) with its layout resolved and every array checked
    against the shapes the format prescribes; needs no GPU.  Returns a dict of numpy arrays plus
    'rows' (the ASCII grid)."""
    if not isinstance(rec, dict):
        with np.load(rec, allow_pickle=False) as z:
            rec = {k: z[k] for k in z.files}
    rec = dict(rec)
    rows = [str(r) for r in rec["layout"]] if "layout" in rec else get_layout(str(rec["layout_name"]))
    if len({len(r) for r in rows}) != 1:
        raise ValueError("layout rows differ in width")
    H, W = len(rows), len(rows[0])
    cells = np.asarray(rec["spawn_cells"])
    if cells.ndim != 3 or cells.shape[2] != 2:
        raise ValueError(f"spawn_cells must be [E, A, 2], got {cells.shape}")
    E, A = cells.shape[:2]
    actions = np.asarray(rec["actions"])
    if actions.ndim != 3 or actions.shape[1:] != (E, A):
        raise ValueError(f"actions must be [K, {E}, {A}], got {actions.shape}")
    K = actions.shape[0]
    if K > int(rec["max_steps"]):
        raise ValueError(f"{K} recorded steps exceed max_steps={int(rec['max_steps'])}: a recording covers one episode")
    if actions.min(initial=0) < 0 or actions.max(initial=0) > 6:
        raise ValueError("actions outside 0..6")
    want = {"spawn_dirs": (E, A), "serve_rank": (K, E, A), "rewards": (K, E, A), "truncated": (K, E, A),
            "agents": (K + 1, E, A, 4), "items": (K + 1, E, H * W), "contested": (K, E)}
    for name, shape in want.items():
        if name in rec and tuple(np.asarray(rec[name]).shape) != shape:
            raise ValueError(f"{name} must have shape {shape}, got {tuple(np.asarray(rec[name]).shape)}")
    for name, lead in (("obs", (K + 1, E, A)), ("pots", (K + 1, E))):
        if name in rec and tuple(np.asarray(rec[name]).shape[:len(lead)]) != lead:
            raise ValueError(f"{name} must start with {lead}, got {tuple(np.asarray(rec[name]).shape)}")
    if "spawn_dirs" not in rec:
        raise ValueError("spawn_dirs is missing")
    if (cells[..., 0] < 0).any() or (cells[..., 0] >= H).any() or (cells[..., 1] < 0).any() or (cells[..., 1] >= W).any():
        raise ValueError("spawn_cells outside the grid")
    rec["rows"] = rows
    return rec


def replay(rec, device=None):
    """Runs the recording; returns {array name: None | (description of the first difference)}
    plus the produced arrays under 'produced'."""
    rec = load_recording(rec)
    rows = rec["rows"]
    feature = str(rec["feature"])
    max_steps = int(rec["max_steps"])
    cells = np.asarray(rec["spawn_cells"])
    E, A = cells.shape[:2]
    actions = np.asarray(rec["actions"])
    K = actions.shape[0]
    W = len(rows[0])
    env = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=max_steps, device=device, auto_reset=False)
    N, D = env.N, env.D
    dev = env.device
    spawn = torch.from_numpy((cells[..., 0] * W + cells[..., 1]).astype(np.uint8).reshape(N)).to(dev)
    dirs = torch.from_numpy(np.asarray(rec["spawn_dirs"]).astype(np.uint8).reshape(N)).to(dev)
    # every step's rows start 16-byte aligned (the kernel's requirement): when N * D is no multiple of
    # four floats, each step's slab is padded to one
    slab = (N * D + 3) // 4 * 4
    obs_store = torch.empty((K + 1, slab), dtype=torch.float32, device=dev)
    obs = [obs_store[k, :N * D].view(N, D) for k in range(K + 1)]
    env.reset(out=obs[0], spawn_override=spawn, dir_override=dirs)
    want_state = any(k in rec for k in ("agents", "pots", "items"))
    states = [env.export_state()] if want_state else []
    act = torch.from_numpy(actions.astype(np.int64).reshape(K, N)).to(dev)
    prio = None
    if "serve_rank" in rec:
        prio = torch.from_numpy(np.asarray(rec["serve_rank"]).astype(np.uint8).reshape(K, N)).to(dev)
    rew = torch.empty((K, N), dtype=torch.float32, device=dev)
    term = torch.empty((K, N), dtype=torch.bool, device=dev)
    trunc = torch.empty((K, N), dtype=torch.bool, device=dev)
    if want_state or (N * D) % 4 != 0:
        for k in range(K):  # snapshots are wanted after every step: step by step
            env.step_into(act[k], obs[k + 1], rew[k], term[k], trunc[k], prio_override=None if prio is None else prio[k])
            if want_state:
                states.append(env.export_state())
    else:
        env.step_many_into(act, obs_store[1:].view(K, N, D), rew, term, trunc, prio_override=prio)
    produced = {"obs": obs_store[:, :N * D].cpu().numpy().reshape(K + 1, E, A, D), "rewards": rew.cpu().numpy().reshape(K, E, A),
                "truncated": trunc.cpu().numpy().reshape(K, E, A)}
    if want_state:
        produced["agents"] = np.stack([s["agents"] for s in states])
        produced["pots"] = np.stack([s["pots"] for s in states])
        produced["items"] = np.stack([s["items"] for s in states])
    report = {"produced": produced}
    for name, got in produced.items():
        if name not in rec:
            continue
        want = np.asarray(rec[name])
        if want.shape != got.shape:
            report[name] = f"shape {want.shape} recorded, {got.shape} produced"
            continue
        bad = np.argwhere(want.astype(got.dtype) != got)
        if bad.size == 0:
            report[name] = None
            continue
        first = tuple(int(x) for x in bad[0])
        what = ""
        if name == "obs":
            what = f" ({feature_at(feature, A, first[3])})"
            first_desc = f"after {'reset' if first[0] == 0 else f'step {first[0] - 1}'}, env {first[1]}, agent {first[2]}, position {first[3]}"
        else:
            first_desc = f"index {first}"
        report[name] = (f"{len(bad)} entries differ; first: {first_desc}{what}: recorded {want[first]}, "
                        f"produced {got[first]}")
        if "contested" in rec and len(first) >= 2:
            # outputs after step k sit at index k + 1 of the [K+1, ...] arrays and at index k of the [K, ...] ones
            upto = first[0] if got.shape[0] == K + 1 else first[0] + 1
            seen = bool(np.asarray(rec["contested"])[:upto, first[1]].any())
            report[name] += ("; a contested step (reconstructed serving order) precedes it in that env" if seen else
                             "; no contested step precedes it in that env: the rules differ, not the serving order")
    return report


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print(__doc__)
        return 2
    report = replay(argv[0])
    failed = False
    for name, diff in report.items():
        if name == "produced":
            continue
        print(f"{name}: {'identical' if diff is None else diff}")
        failed = failed or diff is not None
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())

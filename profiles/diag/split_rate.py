"""Kept-rows regime (every step's rows stay, ring of 32 observation tensors): fused single steps,
the fused multi-step launch, and the split multi-step pipeline (int8 rows + widening kernel) with
different ring depths / store policies.  us per step and TB/s of algorithmic bytes.
    sh profiles/diag/build_variants.sh split:-DOC_SPLIT && MARL_AMD_LIB=$PWD/profiles/diag/libvar_split.so \
    python profiles/diag/split_rate.py [envs] [layout] [agents] [feature]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from marl_amd.env import BatchedOvercookedVecEnv  # noqa: E402
from marl_amd.layouts import get_layout  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
layout = sys.argv[2] if len(sys.argv) > 2 else "overcooked_cramped_room_v0"
A = int(sys.argv[3]) if len(sys.argv) > 3 else 2
feature = sys.argv[4] if len(sys.argv) > 4 else "global_obs"
dev = torch.device("cuda:0")


def run(split, policy="0", ring=32, reps=8, mode="0", graph=False):
    os.environ["MARL_AMD_SPLIT_MODE"] = mode
    os.environ["MARL_AMD_SPLIT"] = str(split)
    os.environ["MARL_AMD_SPLIT_POLICY"] = policy
    env = BatchedOvercookedVecEnv(get_layout(layout), A, feature, E, max_steps=1000, seed=1234, device=dev)
    N, D = env.N, env.D
    ring = int(max(4, min(ring, (6 << 30) // (N * D * 4))))
    B = 4 * D + 6 + 8 + 2 * 4 * env.state_words / A
    gen = torch.Generator(device=dev).manual_seed(1)
    acts = torch.randint(0, 7, (ring, N), dtype=torch.int64, device=dev, generator=gen)
    store = torch.empty((ring, N, D), dtype=torch.float32, device=dev)
    rew = torch.empty((ring, N), dtype=torch.float32, device=dev)
    te = torch.empty((ring, N), dtype=torch.bool, device=dev)
    tr = torch.empty_like(te)
    env.reset(out=store[0])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {}
    if split == 0:
        for i in range(2 * ring):
            env.step_into(acts[i % ring], store[i % ring], rew[0], te[0], tr[0])
        torch.cuda.synchronize(); ev0.record()
        for i in range(reps * ring):
            env.step_into(acts[i % ring], store[i % ring], rew[0], te[0], tr[0])
        ev1.record(); torch.cuda.synchronize()
        out["single"] = ev0.elapsed_time(ev1) * 1e3 / (reps * ring)
    env.step_many_into(acts, store, rew, te, tr)
    torch.cuda.synchronize(); ev0.record()
    for _ in range(reps):
        env.step_many_into(acts, store, rew, te, tr)
    ev1.record(); torch.cuda.synchronize()
    out["many"] = ev0.elapsed_time(ev1) * 1e3 / (reps * ring)
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            env.step_many_into(acts, store, rew, te, tr)
        g.replay()
        torch.cuda.synchronize(); ev0.record()
        for _ in range(reps):
            g.replay()
        ev1.record(); torch.cuda.synchronize()
        out["many, graph replay"] = ev0.elapsed_time(ev1) * 1e3 / (reps * ring)
    return {k: f"{v:.2f} us = {N * B / v / 1e6:.2f} TB/s" for k, v in out.items()}, ring


print(f"{layout} A={A} {feature} E={E}", flush=True)
print("fused:", run(0), flush=True)
print("fused, graph:", run(0, graph=True), flush=True)
for r in (2, 3, 4):
    for pol in ("1", "3"):
        print(f"split two streams ring={r} policy={pol}:", run(r, pol, graph=True), flush=True)
for pol in ("1", "3"):
    print(f"split one stream policy={pol}:", run(3, pol, mode="1", graph=True), flush=True)

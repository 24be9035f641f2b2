"""Halves (or quarters) of the envs as separate env handles on separate streams, rows kept, replayed
from one hipGraph with a branch per stream (launches issued from Python are bound by the host at
two or more launches per step): does one part's simulation hide behind another part's row stream?"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
rows = get_layout("overcooked_cramped_room_v0")
A, total, RING = 2, 65536, 32
def run(parts):
    chunk = total // parts
    side = [torch.cuda.Stream() for _ in range(parts)]
    envs, bufs = [], []
    for k in range(parts):
        e = BatchedOvercookedVecEnv(rows, A, "global_obs", chunk, seed=1, env_index_base=k * chunk)
        n = chunk * A
        b = dict(obs=torch.empty((RING, n, e.D), device="cuda"), rew=torch.empty(n, device="cuda"),
                 term=torch.empty(n, dtype=torch.bool, device="cuda"), trunc=torch.empty(n, dtype=torch.bool, device="cuda"),
                 acts=torch.randint(0, 7, (RING, n), device="cuda"))
        e.reset(out=b["obs"][0])
        envs.append(e); bufs.append(b)
    def branch(k):
        b = bufs[k]
        for i in range(RING):
            envs[k].step_into(b["acts"][i], b["obs"][(i + 1) % RING], b["rew"], b["term"], b["trunc"])
    for k in range(parts):
        branch(k)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for k in range(parts):
            side[k].wait_stream(main)
            with torch.cuda.stream(side[k]):
                branch(k)
        for k in range(parts):
            main.wait_stream(side[k])
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * RING)
    print(f"{parts} branch(es) of {chunk} envs: {us:.2f} us per step of all {total} envs, {total * A * 842 / us / 1e6:.2f} TB/s algorithmic", flush=True)
for parts in (1, 2, 4, 1, 2):
    run(parts)

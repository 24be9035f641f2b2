"""Does the simulation of one half of the envs hide behind the store stream of the other half when
the halves run as separate env handles on separate streams?"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
rows = get_layout("overcooked_cramped_room_v0")
A, total = 2, 65536
RING = int(sys.argv[1]) if len(sys.argv) > 1 else 2  # observation tensors per part (32: rows kept, they drain to HBM)
def run(parts, steps=2000):
    chunk = total // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    envs, bufs = [], []
    for k in range(parts):
        with torch.cuda.stream(streams[k]):
            e = BatchedOvercookedVecEnv(rows, A, "global_obs", chunk, seed=1, env_index_base=k * chunk)
            n = chunk * A
            b = dict(obs=[torch.empty((n, e.D), device="cuda") for _ in range(RING)], rew=torch.empty(n, device="cuda"),
                     term=torch.empty(n, dtype=torch.bool, device="cuda"), trunc=torch.empty(n, dtype=torch.bool, device="cuda"),
                     acts=torch.randint(0, 7, (8, n), device="cuda"))
            e.reset(out=b["obs"][0])
        envs.append(e); bufs.append(b)
    torch.cuda.synchronize()
    def step(i):
        for k in range(parts):
            with torch.cuda.stream(streams[k]):
                b = bufs[k]
                envs[k].step_into(b["acts"][i % 8], b["obs"][i % RING], b["rew"], b["term"], b["trunc"])
    for i in range(100):
        step(i)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) * 1e6 / steps
    print(f"ring {RING}: {parts} stream(s) of {chunk} envs: {us:.2f} us per step of all {total} envs, {total * A * 842 / us / 1e6:.2f} TB/s algorithmic")
for parts in (1, 2, 4, 1, 2):
    run(parts)

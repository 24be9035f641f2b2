"""Env batch cut into shards (env handles of their own) on streams of their own, rows kept, one
hipGraph with a branch per shard and no join between steps.  Matrix over total env count, shard
count and launch form (single-step launches, or one multi-step launch of 32 steps per shard).
    python profiles/diag/shards_matrix.py [layout]"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
layout = sys.argv[1] if len(sys.argv) > 1 else "overcooked_cramped_room_v0"
rows = get_layout(layout)
A, RING = 2, 32

def sleep_cycles_per_us():
    torch.cuda._sleep(1000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(2_000_000); e1.record(); e1.synchronize()
    return 2_000_000 / (e0.elapsed_time(e1) * 1e3)

CPU = sleep_cycles_per_us()
print(f"torch.cuda._sleep: {CPU:.1f} cycles per us", flush=True)

def run(total, parts, many, stagger=False):
    chunk = total // parts
    side = [torch.cuda.Stream() for _ in range(parts)]
    envs, bufs = [], []
    for k in range(parts):
        e = BatchedOvercookedVecEnv(rows, A, "global_obs", chunk, seed=1, env_index_base=k * chunk)
        n = chunk * A
        b = dict(obs=torch.empty((RING, n, e.D), device="cuda"), rew=torch.empty((RING, n), device="cuda"),
                 term=torch.empty((RING, n), dtype=torch.bool, device="cuda"), trunc=torch.empty((RING, n), dtype=torch.bool, device="cuda"),
                 acts=torch.randint(0, 7, (RING, n), device="cuda"))
        e.reset(out=b["obs"][0])
        envs.append(e); bufs.append(b)
    B = 4 * envs[0].D + 4 + 1 + 1 + 8 + 2 * 4 * envs[0].state_words / A
    def branch(k):
        b = bufs[k]
        if many:
            envs[k].step_many_into(b["acts"], b["obs"], b["rew"], b["term"], b["trunc"])
        else:
            for i in range(RING):
                envs[k].step_into(b["acts"][i], b["obs"][(i + 1) % RING], b["rew"][0], b["term"][0], b["trunc"][0])
    for k in range(parts):
        branch(k)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for k in range(parts):
            side[k].wait_stream(main)
            with torch.cuda.stream(side[k]):
                if stagger and k:
                    # shard k starts k / parts of a step late, so that equal-length launches stay out of phase
                    torch.cuda._sleep(int(CPU * total * A * B / 5.4e6 * k / parts))
                branch(k)
        for k in range(parts):
            main.wait_stream(side[k])
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * RING)
    print(f"{layout} {total:>8} envs, {parts} shard(s) of {chunk:>7}, {'multi-step launch' if many else 'single steps     '}: "
          f"{'staggered ' if stagger else ''}{us:7.2f} us per step, {total * A * B / us / 1e6:.2f} TB/s = {total * A * B / us / 1e6 / 8:.3f}", flush=True)
    del g, envs, bufs
    torch.cuda.empty_cache()

for total in (65536, 131072, 262144):
    for parts in (2, 3, 4):
        if total % parts or (total // parts) % 2:
            continue
        for stagger in (False, True):
            run(total, parts, False, stagger)

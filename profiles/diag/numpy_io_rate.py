"""Rate of the host-array compatibility mode (numpy actions in, numpy obs/rewards/flags out)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from marl_amd.env import make_env, make_vector_env, OBS_KEY
for E, reuse in ((16, False), (4096, False), (65536, False), (65536, True)):
    env = make_env(2, "overcooked_cramped_room_v0", "global_obs")
    vec = make_vector_env(E, env, numpy_io=True, seed=1, reuse_host_buffers=reuse)
    vec.reset()
    acts = np.random.default_rng(0).integers(0, 7, size=(8, E * 2))
    n = 200 if E <= 4096 else 30
    for i in range(5): vec.step(acts[i % 8])
    t0 = time.perf_counter()
    for i in range(n): o, r, te, tr, _ = vec.step(acts[i % 8])
    dt = (time.perf_counter() - t0) / n
    print("numpy_io reuse=%s E=%d: %.3f ms per step, %.3e agent-steps/s, obs %s %s" % (reuse, E, dt * 1e3, E * 2 / dt, type(o[OBS_KEY]).__name__, o[OBS_KEY].shape))

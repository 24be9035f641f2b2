"""Four-agent global_obs (D=420) launch time under the tuning knobs: library variant (passes per
wave), waves per workgroup, LDS cap per workgroup.  One process per setting (the knobs are read once)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, json, torch
sys.path.insert(0, %r)
from marl_amd.env import BatchedOvercookedVecEnv
from marl_amd.layouts import get_layout
out = {}
for layout, A, feat, E in (("overcooked_coordination_ring_v0", 4, "global_obs", 32768), ("large_overcooked_layout", 4, "global_obs", 32768),
                           ("overcooked_coordination_ring_v0", 4, "global_obs", 65536), ("overcooked_coordination_ring_v0", 3, "global_obs", 65536)):
    env = BatchedOvercookedVecEnv(get_layout(layout), A, feat, E, seed=1)
    N, D = env.N, env.D
    obs = [torch.empty((N, D), device="cuda") for _ in range(2)]
    rew = torch.empty(N, device="cuda"); te = torch.empty(N, dtype=torch.bool, device="cuda"); tr = torch.empty_like(te)
    acts = torch.randint(0, 7, (16, N), device="cuda")
    env.reset(out=obs[0])
    for k in range(64): env.step_into(acts[k %% 16], obs[k %% 2], rew, te, tr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(512): env.step_into(acts[k %% 16], obs[k %% 2], rew, te, tr)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 512
    B = 4 * D + 6 + 8 + 2 * 4 * env.state_words / A
    out["%%s/%%d/%%d" %% (layout[11:22], A, E)] = [round(us, 2), round(N * B / us / 1e6, 2)]
print(json.dumps(out))
""" % ROOT
for lib in sys.argv[1:]:
    for waves, cap in ((4, 64), (2, 64), (1, 64), (4, 160), (2, 160)):
        env = dict(os.environ, MARL_AMD_WAVES=str(waves), MARL_AMD_LDS_CAP_KB=str(cap))
        if lib != "product":
            env["MARL_AMD_LIB"] = os.path.join(ROOT, "profiles", "diag", lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=120)
        print(lib, "waves", waves, "cap", cap, r.stdout.strip() or r.stderr[-400:], flush=True)

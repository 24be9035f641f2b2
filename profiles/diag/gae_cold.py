"""GAE kernel with inputs that are NOT in the last-level cache: the launches rotate over SETS
distinct (rewards, dones, values) triples of T x N (201 MB each at T=128, N=131,072), so with 4
sets every launch reads memory that 600 MB of other traffic has passed over since its last use --
what an update sees after a rollout.  SETS=1 is the hot case (the same 201 MB every launch).
    python profiles/diag/gae_cold.py [T] [N]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from marl_amd.gae import compute_gae  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
out = {"T": T, "N": N, "algorithmic_bytes": T * N * 20}
for sets in (1, 4):
    data = [(torch.rand(T, N, device="cuda"), (torch.rand(T, N, device="cuda") < 0.01).float(),
             torch.randn(T, N, device="cuda"), torch.randn(1, N, device="cuda")) for _ in range(sets)]
    for k in range(8):
        compute_gae(*data[k % sets], 0.99, 0.95)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    iters = 64
    e0.record()
    for k in range(iters):
        compute_gae(*data[k % sets], 0.99, 0.95)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    out["hot (1 input set)" if sets == 1 else f"cold ({sets} input sets rotated)"] = {
        "us": us, "algorithmic_GBps": T * N * 20 / us / 1e3, "frac_of_8TBps": T * N * 20 / us / 1e3 / 8000}
    del data
print(json.dumps(out, indent=1))

#!/bin/bash
# rocprofv3 evidence for the GAE kernel at T=128, N=131,072: kernel stats, then FETCH_SIZE and
# WRITE_SIZE in separate counter passes.  Usage (GPU box, repo root): bash profiles/gae_profile.sh
set -e
OUT=gpurun_out/prof_gae
mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/gae_run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd.gae import compute_gae
T, N = 128, 131072
r = torch.rand(T, N, device="cuda"); d = (torch.rand(T, N, device="cuda") < 0.01).float()
v = torch.randn(T, N, device="cuda"); nv = torch.randn(1, N, device="cuda")
for _ in range(int(sys.argv[1])):
    compute_gae(r, d, v, nv, 0.99, 0.95)
torch.cuda.synchronize()
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 /tmp/gae_run.py 200 > /dev/null 2> $OUT/trace.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 /tmp/gae_run.py 20 > /dev/null 2> $OUT/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 /tmp/gae_run.py 20 > /dev/null 2> $OUT/write.err
python3 - <<'PY'
import csv, json
def rows(path): return list(csv.DictReader(open(path)))
st = [r for r in rows("gpurun_out/prof_gae/trace/t_kernel_stats.csv") if "gae_kernel" in r["Name"]][0]
f = [float(r["Counter_Value"]) for r in rows("gpurun_out/prof_gae/fetch/f_counter_collection.csv") if "gae_kernel" in r["Kernel_Name"]]
w = [float(r["Counter_Value"]) for r in rows("gpurun_out/prof_gae/write/w_counter_collection.csv") if "gae_kernel" in r["Kernel_Name"]]
alg = 128 * 131072 * 20
out = {"kernel": "gae_kernel, T=128, N=131072", "calls": int(st["Calls"]), "avg_us": float(st["AverageNs"]) / 1e3,
       "algorithmic_bytes": alg, "algorithmic_GBps": alg / float(st["AverageNs"]),
       "FETCH_SIZE_KiB_mean": sum(f) / len(f), "WRITE_SIZE_KiB_mean": sum(w) / len(w),
       "hbm_bytes_per_launch": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024,
       "note": "FETCH_SIZE doubled per the gfx950 correction; counters in KiB; separate passes"}
json.dump(out, open("gpurun_out/prof_gae/gae_profile.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

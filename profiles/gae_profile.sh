#!/bin/bash
# rocprofv3 evidence for the GAE kernel at T=128, N=131,072: kernel stats, then FETCH_SIZE and
# WRITE_SIZE in separate counter passes.  SETS input triples are rotated: 1 = the same 201 MB every
# launch (they stay in the last-level cache), 4 = every launch reads inputs that 600 MB of other
# traffic has passed over (what an update sees after a rollout).
# Usage (GPU box, repo root): SETS=4 bash profiles/gae_profile.sh
set -e
SETS=${SETS:-4}
OUT=gpurun_out/prof_gae_sets$SETS
mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/gae_run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from marl_amd.gae import compute_gae
T, N = 128, 131072
sets = int(sys.argv[2])
data = [(torch.rand(T, N, device="cuda"), (torch.rand(T, N, device="cuda") < 0.01).float(),
         torch.randn(T, N, device="cuda"), torch.randn(1, N, device="cuda")) for _ in range(sets)]
for k in range(int(sys.argv[1])):
    compute_gae(*data[k % sets], 0.99, 0.95)
torch.cuda.synchronize()
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 /tmp/gae_run.py 200 $SETS > /dev/null 2> $OUT/trace.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 /tmp/gae_run.py 20 $SETS > /dev/null 2> $OUT/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 /tmp/gae_run.py 20 $SETS > /dev/null 2> $OUT/write.err
python3 - $OUT $SETS <<'PY'
import csv, json, sys
OUT, SETS = sys.argv[1], int(sys.argv[2])
def rows(path): return list(csv.DictReader(open(path)))
st = [r for r in rows(OUT + "/trace/t_kernel_stats.csv") if "gae_kernel" in r["Name"]][0]
f = [float(r["Counter_Value"]) for r in rows(OUT + "/fetch/f_counter_collection.csv") if "gae_kernel" in r["Kernel_Name"]]
w = [float(r["Counter_Value"]) for r in rows(OUT + "/write/w_counter_collection.csv") if "gae_kernel" in r["Kernel_Name"]]
alg = 128 * 131072 * 20
out = {"kernel": "gae_kernel, T=128, N=131072", "input_sets_rotated": SETS, "calls": int(st["Calls"]), "avg_us": float(st["AverageNs"]) / 1e3,
       "algorithmic_bytes": alg, "algorithmic_GBps": alg / float(st["AverageNs"]),
       "FETCH_SIZE_KiB_mean": sum(f) / len(f), "WRITE_SIZE_KiB_mean": sum(w) / len(w),
       "hbm_bytes_per_launch": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024,
       "note": "FETCH_SIZE doubled per the gfx950 correction; counters in KiB; separate passes"}
json.dump(out, open(OUT + "/gae_profile.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

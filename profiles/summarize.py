"""Condense a profiles/run_profile.sh output directory into the files kept under profiles/<tag>/:
   bench_kernel_stats_<regime>.csv (rocprofv3 --stats rows of our kernels), pmc_traffic.json (HBM
   bytes per launch of the env kernel from the FETCH_SIZE / WRITE_SIZE passes, per regime) and the
   bench lines that ran under the tracer; also refreshes profiles/traffic.json, which bench.py
   quotes.  Usage: python profiles/summarize.py gpurun_out/prof_<tag> profiles/<tag>"""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
here = os.path.dirname(os.path.abspath(__file__))


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"nothing matches {pattern} under {src}")
    return hits[0]


def counter(path, name):
    vals = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == name and "env_kernel" in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return vals


out = {}
traffic_path = os.path.join(here, "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
for regime in ("kept", "llc"):
    with open(find(f"trace_{regime}/**/trace_kernel_stats.csv"), newline="") as f, \
            open(os.path.join(dst, f"bench_kernel_stats_{regime}.csv"), "w", newline="") as g:
        rd = csv.reader(f)
        wr = csv.writer(g)
        wr.writerow(next(rd))
        for row in rd:
            if "oc::" in row[0]:
                wr.writerow(row)
    shutil.copy(os.path.join(src, f"bench_trace_{regime}.json"), os.path.join(dst, f"bench_under_rocprof_{regime}.json"))
    fetch = counter(find(f"pmc_fetch_{regime}/**/fetch_counter_collection.csv"), "FETCH_SIZE")
    write = counter(find(f"pmc_write_{regime}/**/write_counter_collection.csv"), "WRITE_SIZE")
    # step launches only: drop the reset launch (first)
    fetch, write = fetch[1:], write[1:]
    o = {}
    for name, v in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        o[name] = {"launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)}
    # counters are in KiB; on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes: doubled
    o["hbm_bytes_per_launch"] = (2.0 * o["FETCH_SIZE"]["mean_KiB"] + o["WRITE_SIZE"]["mean_KiB"]) * 1024.0
    out[regime] = o
    traffic[f"overcooked_cramped_room_v0/2/global_obs/65536/{regime}"] = int(round(o["hbm_bytes_per_launch"]))
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 64 --warmup 8 "
               "--no-cpu-baseline --no-extra-legs` with --obs-ring 32 (kept) and 2 (llc); counters are KiB; FETCH_SIZE doubled per "
               "the gfx950 correction; env kernel launches only.  The counters sit on the fabric side of the L2s: writes that "
               "the last-level cache absorbs are counted too, so equal numbers in both regimes are expected.")
with open(os.path.join(dst, "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
with open(traffic_path, "w") as f:
    json.dump(traffic, f, indent=1)
print(json.dumps(out, indent=1))

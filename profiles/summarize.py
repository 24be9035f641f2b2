"""Condense a profiles/run_profile.sh output directory into the files kept under profiles/<tag>/:
   bench_kernel_stats.csv (rocprofv3 --stats rows of our kernels), pmc_traffic.json (HBM bytes per
   launch of the env kernel from the FETCH_SIZE / WRITE_SIZE passes) and the bench line that ran
   under the tracer.  Usage: python profiles/summarize.py gpurun_out/prof_<tag> profiles/<tag>"""
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def counter(path, name):
    vals = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == name and "env_kernel" in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return vals


with open(os.path.join(src, "trace", "trace_kernel_stats.csv"), newline="") as f, \
        open(os.path.join(dst, "bench_kernel_stats.csv"), "w", newline="") as g:
    rd = csv.reader(f)
    wr = csv.writer(g)
    wr.writerow(next(rd))
    for row in rd:
        if "oc::" in row[0]:
            wr.writerow(row)
shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, "bench_under_rocprof.json"))

fetch = counter(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write = counter(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
# step launches only: drop the reset launch (first) and anything with a different grid
fetch, write = fetch[1:], write[1:]
out = {}
for name, v in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    out[name] = {"launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)}
# counters are in KiB; on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes: doubled
out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"]["mean_KiB"] + out["WRITE_SIZE"]["mean_KiB"]) * 1024.0
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
               "`bench.py --steps 64 --warmup 8 --no-cpu-baseline`; counters are KiB; FETCH_SIZE doubled per "
               "the gfx950 correction; env kernel launches only")
with open(os.path.join(dst, "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))

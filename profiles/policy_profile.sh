# kernel-time breakdown of the policy-driven bench mode
export TMPDIR=/tmp
OUT=gpurun_out/prof_policy
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 bench.py --mode policy --steps 256 --warmup 32 --no-cpu-baseline --no-rows-kept > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_policy/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over %d kernels" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:24]:
    print("%6.2f%% %6d calls %9.1f us avg  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:120]))
PY

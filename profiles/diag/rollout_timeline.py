"""What one rollout step consists of on the GPU: reads a rocprofv3 kernel trace of a training run,
cuts it at the env kernel's launches and prints, for a typical step, every kernel with its
duration and the idle gap before it.  Usage: python rollout_timeline.py <trace.csv>"""
import csv, sys, statistics
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
cuts = [k for k, r in enumerate(rows) if "env_kernel" in r[2]]
steps = []
for a, b in zip(cuts[:-1], cuts[1:]):
    seg = rows[a:b]
    if 5 <= len(seg) <= 40:  # a rollout step (an update lies between far more kernels)
        steps.append(seg)
steps = steps[len(steps) // 2:]  # the replayed ones
span = [s[-1][1] - s[0][0] for s in steps]
period = [b[0][0] - a[0][0] for a, b in zip(steps[:-1], steps[1:]) if b[0][0] - a[0][0] < 1e6]
busy = [sum(e - s for s, e, _ in seg) for seg in steps]
print("steps %d; kernels per step %s; period %.1f us median; kernel time %.1f us median" % (
    len(steps), statistics.mode(len(s) for s in steps), statistics.median(period) / 1e3, statistics.median(busy) / 1e3))
n = statistics.mode(len(s) for s in steps)
typical = [s for s in steps if len(s) == n]
for k in range(n):
    d = statistics.median(s[k][1] - s[k][0] for s in typical) / 1e3
    g = statistics.median((s[k][0] - s[k - 1][1]) if k else 0 for s in typical) / 1e3
    print("  gap %5.1f us  run %6.1f us  %s" % (g, d, typical[0][k][2][:100]))

"""Kernels of one PPO update in a rocprofv3 kernel trace of a training run: launches, busy time
and idle gaps per kernel name.  Usage: python update_timeline.py <trace.csv>"""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
cuts = [k for k, r in enumerate(rows) if "env_kernel" in r[2]]
segs = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b - a > 100]
a, b = segs[-1]
seg = rows[a + 1:b]
span = (seg[-1][1] - seg[0][0]) / 1e3
busy = sum(e - s for s, e, _ in seg) / 1e3
print("%d updates in the trace; the last one: %d kernels, span %.0f us, busy %.0f us" % (len(segs), len(seg), span, busy))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev = seg[0][0]
for s, e, n in seg:
    k = n[:90]; agg[k][0] += 1; agg[k][1] += (e - s) / 1e3; agg[k][2] += max(0, s - prev) / 1e3; prev = e
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:22]:
    print("%5d calls  run %7.0f us  gaps before %7.0f us  %s" % (v[0], v[1], v[2], k))

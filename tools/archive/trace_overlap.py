#!/usr/bin/env python3
"""Dev tool: summarise a rocprofv3 --kernel-trace CSV: per-kernel stats and how much kernels overlap in time."""
import csv, glob, sys, collections
paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r.get("Queue_Id", 0) or 0), int(r.get("VGPR_Count", 0) or 0)))
rows.sort()
if not rows: sys.exit("no rows")
t0 = rows[0][0]
stat = collections.defaultdict(lambda: [0, 0])
for s, e, n, q, v in rows:
    stat[n][0] += 1; stat[n][1] += e - s
span = rows[-1][1] - t0
busy = 0; cur_end = t0
ev = sorted([(s, 1) for s, e, *_ in rows] + [(e, -1) for s, e, *_ in rows])
depth = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[depth] += t - last; last = t; depth += d
print("span %.3f ms, kernels %d, queues %s" % (span / 1e6, len(rows), sorted(set(r[3] for r in rows))))
print("time by number of concurrently running kernels:", {k: "%.2f ms" % (v / 1e6) for k, v in sorted(hist.items())})
for n, (c, t) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s %5d x %9.1f us avg" % (n[:28], c, t / c / 1e3))
if len(sys.argv) > 2:
    for s, e, n, q, v in rows[int(sys.argv[2]):int(sys.argv[2]) + 60]:
        print("%10.1f %10.1f q%-3d v%-3d %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, v, n[:30]))

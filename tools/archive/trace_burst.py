#!/usr/bin/env python3
"""Dev tool: timeline of the LAST K-step region of `bench.py --steps K` from a rocprofv3 --kernel-trace CSV: for every
launch kind, when the first one starts and the last one ends (relative to the region's first kernel), and per queue the
order of kernels -- shows whether a burst of K batches moves through the launches in lock-step and where the tail is."""
import csv, glob, sys, collections
K = int(sys.argv[2])
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r.get("Queue_Id", 0) or 0)))
rows.sort()
s1 = [r for r in rows if r[2] == "k_rp_stage1"]
t0 = s1[-K][0]
reg = [r for r in rows if r[0] >= t0 and r[2].startswith("k_")]
end = max(r[1] for r in reg)
print("region: %d kernels, %.3f ms" % (len(reg), (end - t0) / 1e6))
by = collections.defaultdict(list)
for r in reg: by[r[2]].append(r)
for n, rs in sorted(by.items(), key=lambda kv: min(x[0] for x in kv[1])):
    print("  %-16s n=%3d first start %7.1f us  last start %7.1f  last end %7.1f   avg dur %7.1f us" % (
        n, len(rs), (min(x[0] for x in rs) - t0) / 1e3, (max(x[0] for x in rs) - t0) / 1e3, (max(x[1] for x in rs) - t0) / 1e3,
        sum(x[1] - x[0] for x in rs) / len(rs) / 1e3))
ev = sorted([(s, 1) for s, e, *_ in reg] + [(e, -1) for s, e, *_ in reg])
depth = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[depth] += t - last; last = t; depth += d
print("  time by concurrently running kernels:", {k: "%.2f ms" % (v / 1e6) for k, v in sorted(hist.items())})
if len(sys.argv) > 3 and sys.argv[3] == "detail":
    for r in sorted(reg):
        print("    q%-3d %-22s %8.1f -> %8.1f us  (%7.1f)" % (r[3], r[2], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3))
qs = collections.Counter(r[3] for r in reg)
print("  queues used: %d" % len(qs), dict(qs))

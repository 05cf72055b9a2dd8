#!/usr/bin/env python3
"""Dev tool: from a rocprofv3 --kernel-trace CSV, per-queue busy fraction, gaps between consecutive kernels of a queue,
and the distribution of the number of concurrently executing kernels (steady-state window only)."""
import csv, glob, sys, collections
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r.get("Queue_Id", 0) or 0)))
rows.sort()
rp = [r for r in rows if "k_rp_stage1" in r[2]]
t0, t1 = rp[len(rp) // 4][0], rp[3 * len(rp) // 4][0]       # middle half of the run
win = [r for r in rows if t0 <= r[0] < t1]
span = t1 - t0
byq = collections.defaultdict(list)
for r in win: byq[r[3]].append(r)
print("window %.2f ms, %d kernels, %d queues" % (span / 1e6, len(win), len(byq)))
busy = []; gaps = collections.defaultdict(list)
for q, rs in sorted(byq.items()):
    b = sum(e - s for s, e, *_ in rs); busy.append(b / span)
    for a, c in zip(rs, rs[1:]): gaps[(a[2][:14], c[2][:14])].append(c[0] - a[1])
print("per-queue busy fraction: min %.2f avg %.2f max %.2f" % (min(busy), sum(busy) / len(busy), max(busy)))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("  gap %-14s -> %-14s n=%4d avg %7.1f us" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
ev = sorted([(s, 1) for s, e, *_ in win] + [(e, -1) for s, e, *_ in win])
depth = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[depth] += t - last; last = t; depth += d
tot = sum(hist.values())
print("concurrency:", {k: "%.0f%%" % (100 * v / tot) for k, v in sorted(hist.items())})
st = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in win: st[n][0] += 1; st[n][1] += e - s
for n, (c, t) in sorted(st.items(), key=lambda kv: -kv[1][1])[:8]:
    print("  %-26s %5d x %8.1f us" % (n[:26], c, t / c / 1e3))

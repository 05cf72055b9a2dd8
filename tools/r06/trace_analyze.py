#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel trace (csv): per queue the share of time a kernel of it is running, the gaps between a
queue's consecutive kernels, and over the whole trace how many kernels run at once.  Usage: trace_analyze.py kernel_trace.csv [t0_frac t1_frac]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?") + "/s" + r.get("Stream_Id", "?"), int(r["Grid_Size_X"]),
                 int(r.get("Workgroup_Size_X", 0) or 0)))
rows.sort()
# window: between the lo-th and hi-th quantile of the start times of the launches of the anchor kernel (name substring, grid)
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_fb_accum"
agrid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lo = float(sys.argv[4]) if len(sys.argv) > 4 else 0.25
hi = float(sys.argv[5]) if len(sys.argv) > 5 else 0.75
st = sorted(r[0] for r in rows if anchor in r[2] and (agrid == 0 or r[4] == agrid))
a, b = st[int(len(st) * lo)], st[int(len(st) * hi)]
sel = [r for r in rows if r[0] >= a and r[1] <= b]
span = (max(r[1] for r in sel) - min(r[0] for r in sel)) / 1e3
print("window: %.1f us, %d kernels, %d queues" % (span, len(sel), len(set(r[3] for r in sel))))
# concurrency histogram
ev = []
for s, e, *_ in sel:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
cur, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[cur] += t - last
    last = t
    cur += d
tot = sum(hist.values())
print("kernels running at once (share of time): " + "  ".join("%d:%.2f" % (k, v / tot) for k, v in sorted(hist.items())))
print("mean kernels in flight: %.2f" % (sum(k * v for k, v in hist.items()) / tot))
# waves in flight (upper bound: every kernel counted with all its waves for its whole duration)
byq = collections.defaultdict(list)
for r in sel:
    byq[r[3]].append(r)
print("per queue: busy share, kernels, median gap between consecutive kernels (us)")
gaps_all = []
for q, v in sorted(byq.items()):
    v.sort()
    busy = sum(e - s for s, e, *_ in v)
    gaps = sorted((v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1))
    gaps_all += gaps
    print("  q%-4s busy %.2f  n %4d  gap p50 %8.1f  p90 %8.1f" % (q, busy / (v[-1][1] - v[0][0]), len(v), gaps[len(gaps) // 2] if gaps else 0, gaps[int(len(gaps) * 0.9)] if gaps else 0))
gaps_all.sort()
print("all gaps: p10 %.1f p50 %.1f p90 %.1f us" % (gaps_all[len(gaps_all) // 10], gaps_all[len(gaps_all) // 2], gaps_all[int(len(gaps_all) * 0.9)]))
byk = collections.defaultdict(list)
for s, e, k, q, g, w in sel:
    byk[(k, g)].append((e - s) / 1e3)
print("kernel (grid): n, median us, total ms")
for (k, g), v in sorted(byk.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("  %-36s %8d  n %4d  p50 %8.1f  total %8.2f" % (k[:36], g, len(v), v[len(v) // 2], sum(v) / 1e3))

#!/usr/bin/env python3
"""Per (kernel, grid size): dispatches and the average of one counter, from a rocprofv3 --pmc run's counter_collection.csv.
Usage: pmc_by_grid.py dir counter [scale]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
ctr = sys.argv[2]
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != ctr: continue
    n = re.sub(r"^void\s+", "", r["Kernel_Name"].split("(")[0])
    k = (n, int(r["Grid_Size"]))
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (n, g), v in sorted(acc.items(), key=lambda kv: -kv[1][1] / kv[1][0]):
    if n.startswith(("k_fb_fill", "k_fb_norm", "k_fb_base", "k_from_uniform", "at::")): continue
    print("%-34s grid %9d  n %4d  avg %16.1f" % (n[:34], g, v[0], v[1] / v[0] * scale))

#!/bin/bash
# round 6, call 1: baseline of the config-5 chain as round 5 left it -- the bench figure on 16 lanes, and the kernel trace of ONE lane
# (every chain alone on the device: per-kernel durations without contention).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --cfg5-only 16 > $OUT/cfg5_16.json 2> $OUT/cfg5_16.err
python $REPO/bench.py --cfg5-only 1 > $OUT/cfg5_1.json 2> $OUT/cfg5_1.err
rm -rf /tmp/pf1
rocprofv3 --kernel-trace --stats -d /tmp/pf1 -o t --output-format csv -- python $REPO/bench.py --cfg5-only 1 > /tmp/pf1.log 2>&1
cp $(find /tmp/pf1 -name "*kernel_stats.csv" | head -1) $OUT/cfg5_1_kernel_stats.csv
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pf1/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    acc[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/cfg5_1_by_grid.txt", "w") as o:
    for (name, grid), v in sorted(acc.items()):
        v.sort()
        o.write("%-40s grid %8d  n %4d  median %9.1f us  min %9.1f  max %9.1f\n" % (name[:40], grid, len(v), v[len(v)//2], v[0], v[-1]))
PY
tail -3 $OUT/cfg5_16.json | cut -c1-1500
cat $OUT/cfg5_1_by_grid.txt

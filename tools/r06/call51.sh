#!/bin/bash
# round 6, call 51: one 20-step burst region of the headline under the kernel trace: what runs when
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call51
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $REPO/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 --repeat 12 > $OUT/bench.log 2>&1
f=$(ls -S $(find /tmp/tr -name '*kernel_trace.csv') | head -1)
python - "$f" > $OUT/burst_timeline.txt <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Stream_Id", "?"), int(r["Grid_Size_X"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# regions: gaps > 300 us between consecutive kernel starts separate bursts; take the regions made only of rp_* kernels with 4+ stage1 launches
regions, cur = [], [rows[0]]
last_end = rows[0][1]
for r in rows[1:]:
    if r[0] - last_end > 200000:
        regions.append(cur); cur = []
    cur.append(r); last_end = max(last_end, r[1])
regions.append(cur)
good = [[r for r in g if "rp_" in r[2] or "finish" in r[2] or "vb_" in r[2]] for g in regions if 3 <= sum(1 for r in g if "rp_stage1" in r[2]) <= 6]
print("rows", len(rows), "names", collections.Counter(r[2][:30] for r in rows).most_common(12))
print("regions:", len(regions), "burst regions:", len(good), "stage1 per region", [sum(1 for r in g if "rp_stage1" in r[2]) for g in regions])
kk = [r for r in rows if ("rp_" in r[2] or "finish" in r[2] or "vb_" in r[2])]
st = [i for i, r in enumerate(kk) if "rp_stage1" in r[2]]
lo, hi = st[10], st[16]
t0 = kk[lo][0]
print("three bursts (two chains of 10 240 proofs each) in the middle of the run:")
for s_, e, k, sid, grid in kk[lo:hi]:
    print("  %8.1f .. %8.1f  (%7.1f us)  stream %-4s grid %8d  %s" % ((s_ - t0) / 1e3, (e - t0) / 1e3, (e - s_) / 1e3, sid, grid, k[:40]))
PY
head -70 $OUT/burst_timeline.txt; tail -3 $OUT/bench.log | cut -c1-300

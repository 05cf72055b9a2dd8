#!/bin/bash
# round 6, call 65: the one-proof chain's kernel timeline on the final tree
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call65
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for mode in "threads 1" "threads 16"; do
rm -rf $OUT/trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- /tmp/combine_rate $INP 0.5 $mode > $OUT/trace.log 2>&1
f=$(ls -S $(find $OUT/trace -name '*kernel_trace.csv') | head -1)
python - "$f" "$mode" >> $OUT/single_call_kernel_timeline.txt <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
rows = rows[len(rows) // 2:]
by = collections.defaultdict(list)
for s, e, k in rows: by[k].append((e - s) / 1e3)
print("final tree, %s: median kernel durations over the second half of the run" % sys.argv[2])
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort(); print("  %-28s n %5d  p50 %7.1f us" % (k[:28], len(v), v[len(v) // 2]))
st = [s for s, e, k in rows if "stage1" in k]
per = sorted((st[i + 1] - st[i]) / 1e3 for i in range(len(st) - 1))
print("  period between chains p50 %.1f us" % per[len(per) // 2])
PY
rm -rf $OUT/trace
done
cat $OUT/single_call_kernel_timeline.txt

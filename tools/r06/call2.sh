#!/bin/bash
# round 6, call 2: the fused bucket chain (bucket2.h) -- GPU parity tests, then config 5 A/B against bucket.h's chain on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call2
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pool_msm.py -x -q -m gpu > $OUT/pytest_msm.txt 2>&1
tail -5 $OUT/pytest_msm.txt
cd /tmp && export TMPDIR=/tmp
for opt in bucket_chain=1 bucket_chain=0 bucket_chain=0,bucket_lanes=128 bucket_chain=1 bucket_chain=0; do
  echo "== $opt" >> $OUT/cfg5_ab.txt
  python $REPO/bench.py --cfg5-only 16 --opt $opt 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']
        print(j['msms_per_s'], 'MSMs/s  batch alone', j['ms_per_batch_one_stream'], 'ms  single', j['ms_single_msm'], 'ms  b2b', j['ms_single_msm_back_to_back'], r['kernels_us'])
    elif 'invalid' in ln or 'Error' in ln: print(ln.strip())
" >> $OUT/cfg5_ab.txt
done
cat $OUT/cfg5_ab.txt
rm -rf /tmp/pf2
rocprofv3 --kernel-trace --stats -d /tmp/pf2 -o t --output-format csv -- python $REPO/bench.py --cfg5-only 1 > /tmp/pf2.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pf2/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0]
    acc[(name, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/cfg5_1_by_grid.txt", "w") as o:
    for (name, grid), v in sorted(acc.items()):
        if name.startswith("k_fb_fill") or name.startswith("k_fb_norm") or "at::" in name: continue
        v.sort()
        o.write("%-40s grid %8d  n %4d  median %9.1f us  min %9.1f  max %9.1f\n" % (name[:40], grid, len(v), v[len(v)//2], v[0], v[-1]))
PY
cat $OUT/cfg5_1_by_grid.txt

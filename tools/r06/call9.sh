#!/bin/bash
# round 6, call 11: the short-chain tail of narrow bucket chains (k_bk2_leaf + k_msm_tail_fast): parity, single-MSM latency A/B, kernel times
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call11
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pool_msm.py tests/test_gpu_ipp.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for opt in bucket_fast_tail=0 bucket_fast_tail=-1 bucket_fast_tail=0 bucket_fast_tail=-1 bucket_fast_tail=1; do
  python $REPO/bench.py --cfg5-only 16 --opt $opt 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('$opt:', j['msms_per_s'], 'MSMs/s  batch alone', j['ms_per_batch_one_stream'], 'ms  single', j['ms_single_msm'], 'ms  b2b', j['ms_single_msm_back_to_back'], r['kernels_us'])" >> $OUT/cfg5_ab.txt
done
cat $OUT/cfg5_ab.txt
rm -rf /tmp/pf9
rocprofv3 --kernel-trace --stats -d /tmp/pf9 -o t --output-format csv -- python $REPO/bench.py --cfg5-only 1 > /tmp/pf9.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pf9/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0]
    acc[(name, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/cfg5_1_by_grid.txt", "w") as o:
    for (name, grid), v in sorted(acc.items()):
        if name.startswith(("k_fb_fill", "k_fb_norm", "k_fb_base", "k_from_uniform")) or "at::" in name: continue
        v.sort()
        o.write("%-40s grid %8d  n %4d  median %9.1f us  min %9.1f  max %9.1f\n" % (name[:40], grid, len(v), v[len(v)//2], v[0], v[-1]))
PY
cat $OUT/cfg5_1_by_grid.txt

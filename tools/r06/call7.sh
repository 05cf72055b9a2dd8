#!/bin/bash
# round 6, call 7: fe_sub_rr in the point operations (all kernels) + the two-buffer (MSM, window) loop: parity, cfg5 A/B, one-lane kernel times, driver-form line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call7
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pool_msm.py tests/test_gpu_rangeproof.py tests/test_gpu_bench_config.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for opt in bucket_two_buffers=0 bucket_two_buffers=1 bucket_two_buffers=0 bucket_two_buffers=1; do
  python $REPO/bench.py --cfg5-only 16 --opt $opt 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('$opt:', j['msms_per_s'], 'MSMs/s  batch alone', j['ms_per_batch_one_stream'], 'ms  single', j['ms_single_msm'], 'ms  b2b', j['ms_single_msm_back_to_back'], r['kernels_us'])" >> $OUT/cfg5_ab.txt
done
cat $OUT/cfg5_ab.txt
for opt in bucket_two_buffers=0 bucket_two_buffers=1; do
rm -rf /tmp/pf7
rocprofv3 --kernel-trace --stats -d /tmp/pf7 -o t --output-format csv -- python $REPO/bench.py --cfg5-only 1 --opt $opt > /tmp/pf7.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pf7/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0]
    acc[(name, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/cfg5_1_by_grid_$opt.txt", "w") as o:
    for (name, grid), v in sorted(acc.items()):
        if name.startswith(("k_fb_fill", "k_fb_norm", "k_fb_base", "k_from_uniform")) or "at::" in name: continue
        v.sort()
        o.write("%-40s grid %8d  n %4d  median %9.1f us  min %9.1f  max %9.1f\n" % (name[:40], grid, len(v), v[len(v)//2], v[0], v[-1]))
PY
echo "== $opt"; cat $OUT/cfg5_1_by_grid_$opt.txt
done
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
j = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
e = j.get("extra", {})
print("headline", j["value"], "ms/step", j["ms_per_step"])
for k in ("cfg3", "cfg4_shape", "cfg5_shape", "rlc_batch4096", "small_table"):
    v = e.get(k, {})
    print(k, {kk: vv for kk, vv in v.items() if isinstance(vv, (int, float))})
d = e.get("drop_in_call_shape", {})
for k, v in d.items():
    if isinstance(v, dict): print(k, v.get("verifications_per_s", v.get("msms_per_s")), v.get("latency_ms"))
PY

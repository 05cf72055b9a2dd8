#!/bin/bash
# round 6, call 5: the fused chain under load -- timeline at 16 / 32 lanes, lane-count sweep, then the whole driver-form bench line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 16 24 32; do
  python $REPO/bench.py --cfg5-only $n 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('lanes $n:', j['msms_per_s'], 'MSMs/s  batch alone', j['ms_per_batch_one_stream'], 'ms  single', j['ms_single_msm'], 'ms  b2b', j['ms_single_msm_back_to_back'], r['kernels_us'])" >> $OUT/cfg5_lanes.txt
done
cat $OUT/cfg5_lanes.txt
rm -rf /tmp/pf5
rocprofv3 --kernel-trace -d /tmp/pf5 -o t --output-format csv -- python $REPO/bench.py --cfg5-only 16 > /tmp/pf5.log 2>&1
f=$(find /tmp/pf5 -name "*kernel_trace.csv" | head -1)
python $REPO/tools/r06/trace_analyze.py $f k_bk2_window 131072 0.3 0.7 > $OUT/timeline_cfg5_16_fused.txt 2>&1
cat $OUT/timeline_cfg5_16_fused.txt
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 6000 $OUT/bench_default.json

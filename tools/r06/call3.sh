#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf3
rocprofv3 --kernel-trace -d /tmp/pf3 -o t --output-format csv -- python $REPO/bench.py --cfg5-only 16 > /tmp/pf3.log 2>&1
f=$(find /tmp/pf3 -name "*kernel_trace.csv" | head -1)
head -2 $f
python $REPO/tools/r06/trace_analyze.py $f k_bk2_window 131072 0.3 0.7 > $OUT/timeline_cfg5_16.txt 2>&1
cat $OUT/timeline_cfg5_16.txt
grep -E '^\{' /tmp/pf3.log | cut -c1-400

#!/bin/bash
# round 6, call 22: generator walk with two line buffers (A/B, same box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call22
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_msm.py -x -q -m gpu -k "config5 or shared" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for opt in fb_walk_two_buffers=0 fb_walk_two_buffers=1 fb_walk_two_buffers=0 fb_walk_two_buffers=1 fb_walk_two_buffers=0 fb_walk_two_buffers=1; do
  python $REPO/bench.py --cfg5-only 16 --opt $opt 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('$opt:', j['msms_per_s'], 'MSMs/s  batch alone', j['ms_per_batch_one_stream'], 'ms  single', j['ms_single_msm'], 'ms', r['kernels_us'])" >> $OUT/cfg5_ab.txt
done
cat $OUT/cfg5_ab.txt

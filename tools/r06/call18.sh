#!/bin/bash
# round 6, call 18: window loop with two record buffers at two wavefronts per SIMD (uncapped) against the capped one-buffer form; walk granularity
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call18
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for opt in bucket_two_buffers=0 bucket_two_buffers=1 bucket_two_buffers=0 bucket_two_buffers=1 fb_walk_waves=512 fb_walk_waves=2048 fb_walk_waves=4096; do
  python $REPO/bench.py --cfg5-only 16 --opt $opt 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('$opt:', j['msms_per_s'], 'MSMs/s  batch alone', j['ms_per_batch_one_stream'], 'ms  single', j['ms_single_msm'], 'ms', r['kernels_us'])" >> $OUT/cfg5_ab.txt
done
cat $OUT/cfg5_ab.txt

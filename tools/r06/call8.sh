#!/bin/bash
# round 6, call 8: config-5 shape against the generator tables' window width (table bytes / TLB reach against additions per term)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for opt in fixed_window_bits=15 fixed_window_bits=14 fixed_window_bits=13 fixed_window_bits=12 fixed_window_bits=11 fixed_window_bits=15; do
  python $REPO/bench.py --cfg5-only 16 --opt $opt 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('$opt:', j['msms_per_s'], 'MSMs/s  batch alone', j['ms_per_batch_one_stream'], 'ms  single', j['ms_single_msm'], 'ms  b2b', j['ms_single_msm_back_to_back'], r['kernels_us'])" >> $OUT/cfg5_window_bits.txt
done
cat $OUT/cfg5_window_bits.txt

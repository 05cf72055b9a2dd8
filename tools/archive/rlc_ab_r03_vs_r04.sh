#!/bin/bash
# A/B of the batch-combined path (batches of 4096) between the round-3 tree (ab/r03) and this tree
mkdir -p gpurun_out/r04
OUT=$PWD/gpurun_out/r04/ab_rlc_r03_vs_r04.txt
: > $OUT
for rep in 1 2; do
for d in ab/r03 .; do
  for st in 12 1; do
    (cd $d; python bench.py --rlc --batch 4096 --streams $st --steps 24 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$d streams=$st value=%.0f' % d['value'], json.dumps(d['roofline'].get('kernels_us')))
") >> $OUT 2>&1
  done
done
done
cat $OUT

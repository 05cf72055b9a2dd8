#!/bin/bash
# round 6, call 24: config 3 in the driver's burst form (20 x 256 proofs of m = 16 from an idle pool) cut into 2 / 3 / 4 / 6 chains
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call24
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for parts in 2 3 4 6 2 3 4; do
  BPGPU_HEAVY_PARTS=$parts python $REPO/bench.py --config cfg3 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('cfg3 burst, parts $parts:', j['value'], 'ms/step', j['ms_per_step'], j['config'].get('schedule', '')[-120:])" >> $OUT/cfg3_burst_parts.txt
done
for parts in 2 4; do
  BPGPU_HEAVY_PARTS=$parts python $REPO/bench.py --config cfg4 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('cfg4 burst, parts $parts:', j['value'], 'ms/step', j['ms_per_step'])" >> $OUT/cfg3_burst_parts.txt
done
cat $OUT/cfg3_burst_parts.txt

#!/bin/bash
# Burst experiment (round 3): T resident proofs verified from an idle device as K batches of S = T/K on K streams.
# Answers: what would coalescing the driver's 20 x 1024 burst into fewer, wider launch chains give?
OUT=${1:-gpurun_out/r03a}; mkdir -p $OUT
run() { python bench.py --no-cpu-baseline --no-extra --no-events "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', '->', round(d['value']), 'per s; ms/region', round(d['ms_per_step']*d['steps'],3), 'regions', d['regions']['count'])"; }
{
echo "== driver form"; run --steps 20 --warmup 5
for cfg in "1024 20 12" "1024 20 20" "2048 10 10" "4096 5 5" "5120 4 4" "10240 2 2" "20480 1 1" "4096 4 4" "8192 2 2" "16384 1 1"; do
  set -- $cfg
  echo "== S=$1 K=$2 streams=$3"; run --batch $1 --steps $2 --warmup $2 --streams $3
done
echo "== default"; run
echo "== one stream"; run --streams 1 --steps 64 --warmup 8
} 2>&1 | tee $OUT/burst_sweep.txt

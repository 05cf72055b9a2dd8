#!/bin/bash
# Round 5: staggered bursts (pool option stagger_chains) on the driver's 20-step forms of configs 2, 3, 4 and in steady state, interleaved A/B.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05g
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
one() {   # cfg steps opts tag
    timeout 200 python bench.py --config $1 --steps $2 --warmup 5 --no-extra --no-cpu-baseline --opt $3 > $OUT/$1_$4_steps$2.json 2> $OUT/$1_$4_steps$2.err
    python3 -c "
import json
try:
    d=json.loads([l for l in open('$OUT/$1_$4_steps$2.json') if l.startswith('{')][-1]); print('$1 $4 steps=$2: %.0f /s  %.4f ms/step' % (d['value'], d['ms_per_step']))
except Exception as e: print('$1 $4 $2 FAILED', e)"
}
for rep in 1 2; do
  for sg in 0 1 2; do
    one cfg3 20 stagger_chains=$sg stagger${sg}_$rep
    one cfg2 20 stagger_chains=$sg stagger${sg}_$rep
  done
done
for sg in 0 1 2; do one cfg4 20 stagger_chains=$sg stagger$sg; done
for sg in 0 2; do one cfg2 640 stagger_chains=$sg stagger$sg; one cfg3 320 stagger_chains=$sg stagger$sg; done
(timeout 300 python -m pytest tests/test_gpu_pool.py -q -k "burst or coalesced" 2>&1 | tail -3)

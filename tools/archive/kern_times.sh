#!/bin/bash
# per-kernel durations (dispatch-attached events) at a few batch sizes, one stream
OUT=${1:-gpurun_out/r03b}; mkdir -p $OUT
run() { python bench.py --no-cpu-baseline --no-extra --events-all "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', '->', round(d['value']), 'per s; ms/region', round(d['ms_per_step']*d['steps'],3), d['roofline']['kernels_us'])"; }
{
run --batch 20480 --steps 1 --warmup 1 --streams 1
run --batch 4096 --steps 1 --warmup 1 --streams 1
run --batch 4096 --steps 5 --warmup 5 --streams 5
run --batch 1024 --steps 1 --warmup 1 --streams 1
run --batch 16384 --steps 1 --warmup 1 --streams 1 --splits 16
run --batch 16384 --steps 1 --warmup 1 --streams 1 --splits 32
run --batch 16384 --steps 1 --warmup 1 --streams 1 --splits 64
run --batch 16384 --steps 1 --warmup 1 --streams 1 --splits 8
} 2>&1 | tee $OUT/kern_times.txt

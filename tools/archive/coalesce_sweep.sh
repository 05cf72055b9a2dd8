#!/bin/bash
# bursts of K x 1024 proofs from idle against the pool's chain width
run() { python bench.py --no-cpu-baseline --no-extra --no-events "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']))"; }
for r in 1 2; do
for k in 4 8 12 20 40; do
for cz in 5120 7168 10240; do
run --steps $k --warmup 5 --coalesce $cz
done
done
done

#!/bin/bash
# burst (20 x 1024 from idle) and steady-state rate against the pool's chain width
run() { python bench.py --no-cpu-baseline --no-extra --no-events "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']))"; }
for r in 1 2; do
for cz in 5120 8192 10240 13653 20480; do
run --steps 20 --warmup 5 --coalesce $cz --opt max_chain_proofs=32768
done
for cz in 5120 10240 16384 21845; do
run --coalesce $cz --opt max_chain_proofs=32768
done
done

#!/bin/bash
# interleaved same-box A/B of a library option through bench.py:  tools/ab_opt.sh "split_stage3=1" [rounds]
OPT=$1; R=${2:-3}
run() { python bench.py --no-cpu-baseline --no-extra --no-events "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']))"; }
for r in $(seq $R); do
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --opt $OPT
done
for r in 1 2; do
run
run --opt $OPT
done

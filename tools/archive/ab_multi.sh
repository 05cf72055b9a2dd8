#!/bin/bash
# interleaved same-box comparison of several option sets through bench.py (default protocol and the 20-step burst):
#   tools/ab_multi.sh ROUNDS "" "a_outside=0" "per_proof_radix=32"
R=$1; shift
run() { python bench.py --no-cpu-baseline --no-extra --no-events "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']))"; }
for r in $(seq $R); do
  for o in "$@"; do
    if [ -z "$o" ]; then run; run --steps 20 --warmup 5; else run --opt $o; run --steps 20 --warmup 5 --opt $o; fi
  done
done

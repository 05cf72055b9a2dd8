#!/bin/bash
# interleaved same-box A/B of several library option sets through bench.py:  tools/ab_opts3.sh "" "split_stage1=1" "split_stage1=2" ...
run() { python bench.py --no-cpu-baseline --no-extra --no-events "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']))"; }
for r in 1 2; do
  for o in "$@"; do
    if [ -z "$o" ]; then run --steps 20 --warmup 5; else run --steps 20 --warmup 5 --opt $o; fi
  done
done
for r in 1 2; do
  for o in "$@"; do
    if [ -z "$o" ]; then run; else run --opt $o; fi
  done
done

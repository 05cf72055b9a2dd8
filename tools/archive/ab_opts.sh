#!/bin/bash
# A/B harness for the GPU box, same library, different bench flags, interleaved ROUNDS times:
#   tools/ab_opts.sh "<common bench args>" "<variant A args>" "<variant B args>" ...
COMMON="$1"; shift
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    python bench.py --no-cpu-baseline $COMMON $v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$v]', round(d['value']), d['roofline']['kernels_us'])"
  done
done

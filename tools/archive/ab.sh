#!/bin/bash
# A/B harness for the GPU box: for each ab/<name>.so given, install it as libbpgpu.so and run the bench
# (interleaved, `ROUNDS` times) so that variants are compared on the same box in the same thermal state.
#   tools/ab.sh "<bench args>" name1 name2 ...
ARGS="$1"; shift
ROUNDS=${ROUNDS:-3}
cp bulletproofs_amd/csrc/libbpgpu.so /tmp/keep.so
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    cp ab/$v.so bulletproofs_amd/csrc/libbpgpu.so
    python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['value']), (d.get('roofline') or {}).get('kernels_us'))"
  done
done
cp /tmp/keep.so bulletproofs_amd/csrc/libbpgpu.so

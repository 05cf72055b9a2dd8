#!/bin/bash
# GPU box: interleaved A/B of ab/<name>.so builds on the two bench forms (the driver's and the default).  tools/r04_ab_generic.sh name1 name2 ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export GPU_MAX_HW_QUEUES=16
mkdir -p $REPO/gpurun_out/r04
OUT=$REPO/gpurun_out/r04/ab_${TAG:-generic}.txt
cp bulletproofs_amd/csrc/libbpgpu.so /tmp/keep.so
B="python $REPO/bench.py --no-cpu-baseline --no-extra"
for r in $(seq ${ROUNDS:-3}); do
  for v in "$@"; do
    cp ab/$v.so bulletproofs_amd/csrc/libbpgpu.so
    for args in "--steps 20 --warmup 5" ""; do
      $B $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', '[$args]', round(d['value']), {k: round(x) for k, x in ((d.get('roofline') or {}).get('kernels_us') or {}).items()})" >> $OUT
    done
  done
done
cp /tmp/keep.so bulletproofs_amd/csrc/libbpgpu.so
cat $OUT

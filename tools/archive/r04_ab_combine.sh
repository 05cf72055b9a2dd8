#!/bin/bash
# GPU box: interleaved A/B of ab/<name>.so builds on the combining queue's rates (tools/combine_rate.py modes)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export GPU_MAX_HW_QUEUES=16
mkdir -p $REPO/gpurun_out/r04
OUT=$REPO/gpurun_out/r04/ab_combine_${TAG:-generic}.txt
cp bulletproofs_amd/csrc/libbpgpu.so /tmp/keep.so
for r in $(seq ${ROUNDS:-2}); do
  for v in "$@"; do
    cp ab/$v.so bulletproofs_amd/csrc/libbpgpu.so
    echo "# $v" >> $OUT
    python tools/combine_rate.py --seconds 2 "threads 1" "threads 64" "threads 256" "tickets 16 128" "big 2 4096" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print('$v', d['mode'], d['threads'], d['arg'], round(d['rate_per_s']), 'p50', d['lat_ms']['p50'], 'p99', d['lat_ms']['p99'], 'ppc', d['proofs_per_chain'], 'mism', d['mismatches'])" >> $OUT
  done
done
cp /tmp/keep.so bulletproofs_amd/csrc/libbpgpu.so
cat $OUT

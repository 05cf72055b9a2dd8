#!/bin/bash
# round 6, call 60: host-pointer calls of 4096 proofs from two threads: how the call is sliced (slice_proofs)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call60
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for rep in 1 2; do for o in "none=0" "slice_proofs=1024" "slice_proofs=2048" "slice_proofs=4096" "slice_proofs=8192"; do for mode in "big 2 4096" "big 4 4096" "big 1 16384"; do
  oo=$o; [ $o = "none=0" ] && oo=""
  BP_OPTS=$oo /tmp/combine_rate $INP 1.0 $mode 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$o', '$mode', 'rate', d['rate_per_s'], 'p50', d['lat_ms']['p50'], 'p99', d['lat_ms']['p99'], 'per chain', d['proofs_per_chain'], d['mismatches'], d['errors'])" >> $OUT/slice_sweep.txt
done; done; done
cat $OUT/slice_sweep.txt

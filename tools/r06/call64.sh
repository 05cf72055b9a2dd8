#!/bin/bash
# round 6, call 64: the service thread polling when a narrow chain is due (A/B: BPGPU_NO_DUE_POLL)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call64
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for rep in 1 2 3 4; do for e in "BPGPU_X=1" "BPGPU_NO_DUE_POLL=1"; do for mode in "threads 1" "threads 2" "threads 4"; do
  env $e /tmp/combine_rate $INP 1.0 $mode 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$e', '$mode', 'rate', d['rate_per_s'], 'p50', d['lat_ms']['p50'], 'p99', d['lat_ms']['p99'], d['mismatches'], d['errors'])" >> $OUT/due_poll_ab.txt
done; done; done
cat $OUT/due_poll_ab.txt

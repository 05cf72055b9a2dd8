#!/bin/bash
# round 6, call 57: the sealing deadlines of the combining queue re-swept on the shorter narrow chain (combine_wait_us x combine_quiet_us, combine_inflight)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call57
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for rep in 1 2; do
for o in "combine_wait_us=100,combine_quiet_us=20" "combine_wait_us=60,combine_quiet_us=20" "combine_wait_us=60,combine_quiet_us=10" "combine_wait_us=40,combine_quiet_us=10" "combine_wait_us=150,combine_quiet_us=30" "combine_wait_us=100,combine_quiet_us=20,combine_inflight=8" "combine_wait_us=100,combine_quiet_us=20,combine_inflight=4"; do
  for mode in "threads 16" "threads 64" "threads 256" "tickets 16 128"; do
    BP_OPTS=$o /tmp/combine_rate $INP 1.0 $mode 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$o', '$mode', 'rate', d['rate_per_s'], 'p50', d['lat_ms']['p50'], 'p99', d['lat_ms']['p99'], 'per chain', d['proofs_per_chain'], d['mismatches'], d['errors'])" >> $OUT/seal_sweep.txt
  done
done
done
cat $OUT/seal_sweep.txt

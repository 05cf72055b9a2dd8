#!/bin/bash
# round 6, call 49: the call-shape rows with 8 / 12 / 16 pool lanes (the client's choice: BP_LANES)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call49
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_W=16 GPU_MAX_HW_QUEUES=16
for rep in 1 2; do for lanes in 8 12 16; do for mode in "threads 64" "threads 256" "tickets 16 128" "tickets 16 512"; do
  BP_LANES=$lanes /tmp/combine_rate $INP 1.0 $mode 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('lanes $lanes', '$mode', 'rate', d['rate_per_s'], 'lat', d['lat_ms'], 'per chain', d['proofs_per_chain'], 'mismatches', d['mismatches'], 'errors', d['errors'])" >> $OUT/lanes_ab.txt
done; done; done
cat $OUT/lanes_ab.txt

#!/bin/bash
# round 6, call 46: where the ~9 ms maximum of the call-shape rows comes from (outlier report of tools/combine_rate.cpp), four table levels on / off
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call46
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for e in "BPGPU_NARROW_HI4_MAX=0" "BPGPU_NARROW_HI4_MAX=4" "BPGPU_NARROW_HI4_MAX=0 BPGPU_NARROW_HI_MAX=0"; do
  for mode in "threads 1" "threads 16"; do
    echo "== $e $mode" >> $OUT/outliers.txt
    env $e /tmp/combine_rate $INP 1.0 $mode 2>> $OUT/outliers.txt | grep '^{' | tail -1 | cut -c1-230 >> $OUT/outliers.txt
  done
done
cat $OUT/outliers.txt | cut -c1-600

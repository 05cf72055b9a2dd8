#!/bin/bash
# round 6, call 16: the call-shape rows' outliers against the container's CPU-bandwidth throttling counters
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call16
mkdir -p $OUT
cd $REPO
( echo "quota_us $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) period_us $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null) cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) nproc $(nproc)"; cat /sys/fs/cgroup/cpu/cpu.stat /sys/fs/cgroup/cpu.stat 2>/dev/null ) > $OUT/outliers.txt
g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -o /tmp/cr -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,$REPO/bulletproofs_amd/csrc 2>&1 | tail -3
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for mode in "threads 1" "threads 64" "threads 256" "tickets 16 128" "big 2 4096"; do
  echo "== $mode" >> $OUT/outliers.txt
  /tmp/cr bench_data/combine_rate_inputs.bin 2.0 $mode >> $OUT/outliers.txt 2>&1
done
cut -c1-1200 $OUT/outliers.txt | grep -v '^{"mode"'

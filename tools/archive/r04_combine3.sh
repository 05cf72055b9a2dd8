#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_combine.py tests/test_gpu_msm.py tests/test_gpu_pool.py -x -q > gpurun_out/r04/test_combine.txt 2>&1
echo "tests rc=$?" ; tail -5 gpurun_out/r04/test_combine.txt
R=gpurun_out/r04/combine_rate3.txt
: > $R
run() { echo "# $*" >> $R; timeout 300 python tools/combine_rate.py --seconds 2.5 "$@" >> $R 2>> gpurun_out/r04/combine_rate3.err; }
run "threads 1" "threads 16" "threads 64" "threads 256" "threads 1024" "threads 64 16" "tickets 4 256" "tickets 16 128" "tickets 16 512" "big 1 4096" "big 2 4096" "big 4 4096" "big 1 65536"
run --opts combine_inflight=3 "threads 256" "tickets 16 128" "tickets 16 512"
run --opts combine_inflight=10 "threads 256" "tickets 16 128" "tickets 16 512"
run --opts combine_wait_us=200,combine_quiet_us=40 "threads 64" "threads 256"
cat $R; tail -5 gpurun_out/r04/combine_rate3.err

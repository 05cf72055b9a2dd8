#!/bin/bash
# round 6, call 15: where the latency maxima of the call-shape rows sit (outlier report of tools/combine_rate.cpp), the option-flip test, the table curve
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call15
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_pool.py tests/test_gpu_combine.py tests/test_gpu_concurrency.py tests/test_gpu_pool_msm.py tests/test_gpu_transcript_stop.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -o /tmp/cr -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,$REPO/bulletproofs_amd/csrc 2>&1 | tail -3
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for mode in "threads 1" "threads 64" "threads 256" "tickets 16 128" "big 2 4096"; do
  echo "== $mode" >> $OUT/outliers.txt
  BP_TRACE=/tmp/trace_$(echo $mode | tr ' ' '_').txt /tmp/cr bench_data/combine_rate_inputs.bin 2.0 $mode >> $OUT/outliers.txt 2>&1
done
cut -c1-900 $OUT/outliers.txt

#!/bin/bash
# round 4, first GPU pass: parity of the combining queue, then its rates
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_combine.py -x -q > gpurun_out/r04/test_combine.txt 2>&1
echo "combine tests rc=$?" ; tail -15 gpurun_out/r04/test_combine.txt
timeout 900 python -m pytest tests/test_gpu_pool.py tests/test_gpu_transcripts.py -x -q > gpurun_out/r04/test_pool.txt 2>&1
echo "pool tests rc=$?" ; tail -5 gpurun_out/r04/test_pool.txt
timeout 600 python tools/combine_rate.py --seconds 3 "threads 1" "threads 16" "threads 64" "threads 256" "threads 1024" "threads 64 16" "tickets 4 256" "tickets 16 128" "big 1 4096" "big 2 4096" "big 4 4096" "big 1 65536" > gpurun_out/r04/combine_rate.txt 2> gpurun_out/r04/combine_rate.err
echo "combine_rate rc=$?"; cat gpurun_out/r04/combine_rate.txt; tail -5 gpurun_out/r04/combine_rate.err
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_steps20.json 2> gpurun_out/r04/bench_steps20.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/r04/bench_steps20.json

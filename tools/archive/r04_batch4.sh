#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=16
timeout 1200 python -m pytest tests/test_gpu_pool.py tests/test_gpu_reference_api.py tests/test_gpu_combine.py tests/test_gpu_distributed.py -x -q > gpurun_out/r04/test_batch4.txt 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r04/test_batch4.txt
timeout 1500 python tools/r04_dryrun8.py > gpurun_out/r04/dryrun8.txt 2> gpurun_out/r04/dryrun8.err
echo "dryrun rc=$?"; cat gpurun_out/r04/dryrun8.txt; tail -3 gpurun_out/r04/dryrun8.err

#!/bin/bash
# round 6, call 63: last sanity on the final tree: pool / combine / narrow-chain suites + smoke
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call63
mkdir -p $OUT
cd $REPO
timeout 2000 python -m pytest tests/test_gpu_pool.py tests/test_gpu_combine.py tests/test_gpu_narrow_chain.py tests/test_gpu_bench_config.py tests/test_gpu_msm.py -q -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

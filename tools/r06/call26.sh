#!/bin/bash
# round 6, call 26: the cold-start warm-up of the pooled MSM class (first bpgpu_pool_msm_* call of a thread) + baseline of the call-shape rows
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call26
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_pool_msm.py tests/test_gpu_pool.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
python - <<'PY' > $OUT/call_shape.txt 2>&1
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
r = bench.drop_in_call_shape(False, 0)
for k, v in r.items():
    if isinstance(v, dict):
        print(k, json.dumps(v)[:900])
PY
cat $OUT/call_shape.txt

#!/bin/bash
# round 6, call 19: 64 threads of one 6179-term MSM each -- chains in flight under the cohort policy (constant `cohort_inflight`), default tables
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call19
mkdir -p $OUT
cd $REPO
g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -o /tmp/cr -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,$REPO/bulletproofs_amd/csrc 2>&1 | tail -3
python - <<PY
import sys, os
sys.path.insert(0, "tools")
import make_msm_inputs
make_msm_inputs.write("/tmp/msm_in.bin", 0)
PY
export BP_LANES=8 GPU_MAX_HW_QUEUES=16 BP_MSM_INPUTS=/tmp/msm_in.bin
for w in 12 0; do for tune in cohort_inflight=2 cohort_inflight=3 cohort_inflight=4 cohort_inflight=3,regroup_us=120 cohort_inflight=2; do
  echo "== BP_W=$w $tune" >> $OUT/msm64.txt
  BP_W=$w BP_TUNE=$tune /tmp/cr bench_data/combine_rate_inputs.bin 2.0 msm 64 1 2>/dev/null | cut -c1-330 >> $OUT/msm64.txt
done; done
cat $OUT/msm64.txt

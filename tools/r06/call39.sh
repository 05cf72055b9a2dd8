#!/bin/bash
# round 6, call 39: whole GPU suite (with the aggregated narrow-chain test); the pooled MSM rows' queue timeline and kernel stats
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call39
mkdir -p $OUT
cd $REPO
timeout 3000 python -m pytest tests -q -m gpu > $OUT/pytest_gpu_full.txt 2>&1; tail -5 $OUT/pytest_gpu_full.txt
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=0 GPU_MAX_HW_QUEUES=16
python3 $REPO/tools/make_msm_inputs.py /tmp/msm_inputs.bin > /dev/null 2>&1 && export BP_MSM_INPUTS=/tmp/msm_inputs.bin
for mode in "msm 1 1" "msm 64 1"; do
  name=$(echo $mode | tr ' ' '_')
  BP_TRACE=$OUT/trace_$name.jsonl /tmp/combine_rate $INP 1.0 $mode > $OUT/rate_$name.json 2>/dev/null
  python $REPO/tools/combine_timeline.py $OUT/trace_$name.jsonl > $OUT/timeline_$name.txt 2>&1; rm -f $OUT/trace_$name.jsonl
  rm -rf /tmp/pf_$name
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -o t --output-format csv -- /tmp/combine_rate $INP 1.0 $mode > /tmp/pf_$name.log 2>&1
  cp $(find /tmp/pf_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
done
cat $OUT/timeline_msm_64_1.txt; cut -c1-250 $OUT/rate_msm_64_1.json | tail -2

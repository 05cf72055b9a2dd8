#!/bin/bash
# The combining queue on the GPU box: every call shape of tools/combine_rate.cpp, the ticket regime under a sweep of the sealing
# policy's options, the queue's timeline (BP_TRACE) for the default setting.  Writes gpurun_out/r05_combine/*.
#   bash tools/combine_sweep.sh [seconds per run]
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_combine
mkdir -p $OUT
SECS=${1:-1.5}
export GPU_MAX_HW_QUEUES=16 BP_LANES=8 BP_W=16
g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,$PWD/bulletproofs_amd/csrc -o /tmp/combine_rate || exit 1
INP=bench_data/combine_rate_inputs.bin
run() {   # name, env assignments..., -- args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    echo "== $name: ${envs[*]:-} $*" >> $OUT/log.txt
    env "${envs[@]}" timeout 120 /tmp/combine_rate $INP $SECS "$@" > $OUT/$name.json 2>> $OUT/log.txt
    python3 - "$name" "$OUT/$name.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("%-34s %9.0f /s  p50 %.3f p99 %.3f ms  %7.1f per chain  mism %d err %d  svc issue/complete %s" % (sys.argv[1], d["rate_per_s"], d["lat_ms"]["p50"], d["lat_ms"]["p99"],
          d["proofs_per_chain"], d["mismatches"], d["errors"], d.get("svc")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run threads_1 -- threads 1
run threads_1_unmapped BP_OPTS=combine_mapped_out=0 -- threads 1
run threads_1_b8 -- threads 1 8
run threads_16 -- threads 16
run threads_64 -- threads 64
run threads_256 -- threads 256
run tickets_16x128 BP_TRACE=$OUT/timeline_tickets_16x128.jsonl -- tickets 16 128
run tickets_16x128_wide128 BP_OPTS=combine_wide_proofs=128 -- tickets 16 128
run tickets_16x128_wide128_c2 BP_OPTS=combine_wide_proofs=128,combine_inflight_wide=2 -- tickets 16 128
run tickets_16x128_wide128_c4 BP_OPTS=combine_wide_proofs=128,combine_inflight_wide=4 -- tickets 16 128
run tickets_16x128_wide64_c3_hold800 BP_OPTS=combine_wide_proofs=64,combine_inflight_wide=3,combine_hold_us=800 -- tickets 16 128
run tickets_16x128_inflight3 BP_OPTS=combine_inflight=3 -- tickets 16 128
run tickets_16x128_inflight2_wait300 BP_OPTS=combine_inflight=2,combine_wait_us=300 -- tickets 16 128
run tickets_16x128_unmapped BP_OPTS=combine_mapped_out=0 -- tickets 16 128
run tickets_16x128_lanes24 BPGPU_COMBINE_LANES=24 -- tickets 16 128
run tickets_16x512 -- tickets 16 512
run tickets_16x512_wide128_c3 BP_OPTS=combine_wide_proofs=128 -- tickets 16 512
run tickets_4x512 -- tickets 4 512
run big_2x4096 -- big 2 4096
python3 tools/make_msm_inputs.py /tmp/msm_inputs.bin > /dev/null 2>> $OUT/log.txt
export BP_MSM_INPUTS=/tmp/msm_inputs.bin
run msm_1 BP_W=10 -- msm 1 1
run msm_16 BP_W=10 -- msm 16 1
run msm_64 BP_W=10 BP_TRACE=$OUT/timeline_msm_64.jsonl -- msm 64 1
run msm_64_b2 BP_W=10 -- msm 64 2
python3 tools/combine_timeline.py $OUT/timeline_tickets_16x128.jsonl > $OUT/timeline_tickets_16x128.txt 2>&1
python3 tools/combine_timeline.py $OUT/timeline_msm_64.jsonl > $OUT/timeline_msm_64.txt 2>&1
cat $OUT/timeline_tickets_16x128.txt

#!/bin/bash
# Round 5, second GPU call: the new parity tests, the sealing policies side by side (combine_policy 0 = two regimes, 1 = cohorts) on every call
# shape of tools/combine_rate.cpp, chains planned by proofs vs by work on BASELINE configs 3 / 4 (20-step and 640-step forms) and on the
# mixed-shape figure.  Writes gpurun_out/r05b/*.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05b
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
(timeout 600 python -m pytest tests/test_gpu_transcript_stop.py tests/test_gpu_coalesce_shapes.py tests/test_gpu_combine.py tests/test_gpu_pool.py tests/test_gpu_rlc.py tests/test_gpu_mixed_shapes.py tests/test_gpu_transcripts.py -x -q 2>&1 | tail -25) > $OUT/new_tests.txt
tail -3 $OUT/new_tests.txt
g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,$PWD/bulletproofs_amd/csrc -o /tmp/combine_rate || exit 1
INP=bench_data/combine_rate_inputs.bin
run() {   # name, env assignments..., -- args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    echo "== $name: ${envs[*]:-} $*" >> $OUT/log.txt
    env BP_LANES=8 BP_W=16 "${envs[@]}" timeout 60 /tmp/combine_rate $INP 1.5 "$@" > $OUT/$name.json 2>> $OUT/log.txt
    echo "   rc=$?" >> $OUT/log.txt
    python3 - "$name" "$OUT/$name.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("%-28s %9.0f /s  p50 %.3f p99 %.3f ms  %7.1f per chain  mism %d err %d" % (sys.argv[1], d["rate_per_s"], d["lat_ms"]["p50"], d["lat_ms"]["p99"],
          d["proofs_per_chain"], d["mismatches"], d["errors"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
python3 tools/make_msm_inputs.py /tmp/msm_inputs.bin > /dev/null 2>> $OUT/log.txt
for pol in 0 1; do
    P="BP_OPTS=combine_policy=$pol"
    run p${pol}_threads_1 $P -- threads 1
    run p${pol}_threads_16 $P -- threads 16
    run p${pol}_threads_64 $P -- threads 64
    run p${pol}_threads_256 $P -- threads 256
    run p${pol}_tickets_16x128 $P -- tickets 16 128
    run p${pol}_tickets_16x512 $P -- tickets 16 512
    run p${pol}_tickets_4x512 $P -- tickets 4 512
    run p${pol}_msm_64 $P BP_W=10 BP_MSM_INPUTS=/tmp/msm_inputs.bin -- msm 64 1
done
run p1_c1_threads_64 BP_OPTS=combine_policy=1,combine_cohort_inflight=1 -- threads 64
run p1_c3_threads_256 BP_OPTS=combine_policy=1,combine_cohort_inflight=3 -- threads 256
run p1_c3_tickets_16x128 BP_OPTS=combine_policy=1,combine_cohort_inflight=3 -- tickets 16 128
run p1_c1_msm_64 BP_OPTS=combine_policy=1,combine_cohort_inflight=1 BP_W=10 BP_MSM_INPUTS=/tmp/msm_inputs.bin -- msm 64 1
grep -B2 -A30 WATCHDOG $OUT/log.txt | head -80
# chains by proofs vs by work
for cfg in cfg3 cfg4; do
    for pw in 0 1; do
        for st in 20 320; do
            timeout 200 python bench.py --config $cfg --steps $st --warmup 5 --no-extra --no-cpu-baseline --opt plan_by_work=$pw > $OUT/${cfg}_work${pw}_steps${st}.json 2> $OUT/${cfg}_work${pw}_steps${st}.err
            python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$OUT/${cfg}_work${pw}_steps${st}.json') if l.startswith('{')][-1]); print('$cfg plan_by_work=$pw steps=$st: %.0f /s  %.3f ms/step' % (d['value'], d['ms_per_step']))
except Exception as e: print('$cfg $pw $st FAILED', e)"
        done
    done
done

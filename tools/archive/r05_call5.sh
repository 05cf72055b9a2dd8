#!/bin/bash
# Round 5, fifth GPU call: pooled MSM chains reading their inputs in place (combine_mapped_in) against the staging copy; config 5 without the
# second stream per context.  Writes gpurun_out/r05e/*.
set -u
cd "$(dirname "$0")/../.."
REPO=$PWD
OUT=$REPO/gpurun_out/r05e
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
(timeout 600 python -m pytest tests/test_gpu_pool_msm.py tests/test_gpu_msm.py -q 2>&1 | tail -15) > $OUT/msm_tests.txt
tail -3 $OUT/msm_tests.txt
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
M="BP_W=12 BP_MSM_INPUTS=/tmp/msm_inputs.bin"
for rep in 1 2; do
    for mi in 0 1; do
        run msm_1_in${mi}_$rep $M BP_OPTS=combine_mapped_in=$mi -- msm 1 1
        run msm_16_in${mi}_$rep $M BP_OPTS=combine_mapped_in=$mi -- msm 16 1
        run msm_64_in${mi}_$rep $M BP_OPTS=combine_mapped_in=$mi -- msm 64 1
    done
done
run msm_64_in1_c3 $M BP_OPTS=combine_mapped_in=1,combine_cohort_inflight=3 -- msm 64 1
run msm_64_in1_c4 $M BP_OPTS=combine_mapped_in=1,combine_cohort_inflight=4 -- msm 64 1
run msm_128_in1 $M BP_OPTS=combine_mapped_in=1 -- msm 128 1
run msm_64_b4_in1 $M BP_OPTS=combine_mapped_in=1 -- msm 64 4
run msm_64_in1_trace $M BP_OPTS=combine_mapped_in=1 BP_TRACE=$OUT/timeline_msm_64.jsonl -- msm 64 1
python3 tools/combine_timeline.py $OUT/timeline_msm_64.jsonl > $OUT/timeline_msm_64.txt 2>&1; tail -14 $OUT/timeline_msm_64.txt
grep -B2 -A30 WATCHDOG $OUT/log.txt | head -60
for ns in 16 24 32; do
    for fk in 1 0; do
        timeout 200 python bench.py --cfg5-only $ns --opt msm_fork=$fk > $OUT/cfg5_streams${ns}_fork$fk.json 2> $OUT/cfg5_streams${ns}_fork$fk.err
        python3 -c "
import json
try:
    d=json.loads([l for l in open('$OUT/cfg5_streams${ns}_fork$fk.json') if l.startswith('{')][-1]); print('cfg5 streams=$ns msm_fork=$fk: %.0f MSMs/s single %.3f ms batch alone %.3f ms' % (d['msms_per_s'], d['ms_single_msm'], d['ms_per_batch_one_stream']))
except Exception as e: print('cfg5 $ns $fk FAILED', e)"
    done
done

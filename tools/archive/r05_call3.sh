#!/bin/bash
# Round 5, third GPU call: parity tests of the round (fixed), the per-kind sealing policy, floors for chains planned by work, cfg5 on more streams,
# a kernel trace of one blocking caller.  Writes gpurun_out/r05c/*.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05c
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
(timeout 900 python -m pytest tests/test_gpu_transcript_stop.py tests/test_gpu_coalesce_shapes.py tests/test_gpu_combine.py tests/test_gpu_pool.py tests/test_gpu_rlc.py tests/test_gpu_mixed_shapes.py tests/test_gpu_transcripts.py tests/test_gpu_pool_msm.py -q 2>&1 | tail -40) > $OUT/new_tests.txt
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
run p2_threads_1 -- threads 1
run p2_threads_16 -- threads 16
run p2_threads_64 -- threads 64
run p2_threads_256 -- threads 256
run p2_tickets_16x128 -- tickets 16 128
run p2_tickets_16x512 -- tickets 16 512
run p2_msm_64 BP_W=12 BP_MSM_INPUTS=/tmp/msm_inputs.bin -- msm 64 1
run p2_msm_1 BP_W=12 BP_MSM_INPUTS=/tmp/msm_inputs.bin -- msm 1 1
run p2_inflight8_threads_256 BP_OPTS=combine_inflight=8 -- threads 256
run p2_inflight4_threads_64 BP_OPTS=combine_inflight=4 -- threads 64
grep -B2 -A30 WATCHDOG $OUT/log.txt | head -80
# one blocking caller under the kernel trace: where the 0.53 ms of a call are
(cd /tmp && export TMPDIR=/tmp && BP_LANES=8 BP_W=16 timeout 120 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_threads1 -o t1 -- /tmp/combine_rate $OLDPWD/$INP 0.5 threads 1 > $OLDPWD/$OUT/prof_threads1.json 2> $OLDPWD/$OUT/prof_threads1.err)
find $OUT/prof_threads1 -name "*kernel_stats.csv" | head -1 | xargs -r head -20 | cut -c1-160
# chains by work with a floor (proofs per chain of an aggregated shape)
one() {   # cfg steps opts tag
    timeout 200 python bench.py --config $1 --steps $2 --warmup 5 --no-extra --no-cpu-baseline --opt $3 > $OUT/$1_$4_steps$2.json 2> $OUT/$1_$4_steps$2.err
    python3 -c "
import json
try:
    d=json.loads([l for l in open('$OUT/$1_$4_steps$2.json') if l.startswith('{')][-1]); print('$1 $4 steps=$2: %.0f /s  %.3f ms/step' % (d['value'], d['ms_per_step']))
except Exception as e: print('$1 $4 $2 FAILED', e)"
}
for cfg in cfg3 cfg4; do
    one $cfg 20 plan_by_work=0 proofs
    for fl in 1280 1792 2560; do one $cfg 20 plan_by_work=1,plan_min_chain_proofs=$fl floor$fl; done
    one $cfg 320 plan_by_work=1,plan_min_chain_proofs=1792 floor1792
    one $cfg 320 plan_by_work=1,plan_min_chain_proofs=2560 floor2560
done
# cfg5 shape on more streams
for ns in 8 12 16 24; do
    timeout 200 python bench.py --cfg5-only $ns > $OUT/cfg5_streams$ns.json 2> $OUT/cfg5_streams$ns.err
    python3 -c "
import json
try:
    d=json.loads([l for l in open('$OUT/cfg5_streams$ns.json') if l.startswith('{')][-1]); print('cfg5 streams=$ns: %.0f MSMs/s single %.3f ms batch alone %.3f ms' % (d['msms_per_s'], d['ms_single_msm'], d['ms_per_batch_one_stream']))
except Exception as e: print('cfg5 $ns FAILED', e)"
done

#!/bin/bash
# Round 5, fourth GPU call: the transcript-stop tests (fixed), cohort width for the MSM kinds, the default chain plan against the plan by proofs,
# counters of launch 1 alone (issue / wait / active split) and of config 5's chain (with k_bk_heavy in it).  Writes gpurun_out/r05d/*.
set -u
cd "$(dirname "$0")/../.."
REPO=$PWD
OUT=$REPO/gpurun_out/r05d
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
(timeout 600 python -m pytest tests/test_gpu_transcript_stop.py tests/test_gpu_coalesce_shapes.py -q 2>&1 | tail -40) > $OUT/new_tests.txt
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
for c in 2 3 4 6; do run msm_64_cohorts$c BP_W=12 BP_MSM_INPUTS=/tmp/msm_inputs.bin BP_OPTS=combine_cohort_inflight=$c -- msm 64 1; done
run msm_16 BP_W=12 BP_MSM_INPUTS=/tmp/msm_inputs.bin -- msm 16 1
run msm_64_b4 BP_W=12 BP_MSM_INPUTS=/tmp/msm_inputs.bin -- msm 64 4
grep -B2 -A30 WATCHDOG $OUT/log.txt | head -60
one() {   # cfg steps opts tag
    timeout 200 python bench.py --config $1 --steps $2 --warmup 5 --no-extra --no-cpu-baseline --opt $3 > $OUT/$1_$4_steps$2.json 2> $OUT/$1_$4_steps$2.err
    python3 -c "
import json
try:
    d=json.loads([l for l in open('$OUT/$1_$4_steps$2.json') if l.startswith('{')][-1]); print('$1 $4 steps=$2: %.0f /s  %.3f ms/step' % (d['value'], d['ms_per_step']))
except Exception as e: print('$1 $4 $2 FAILED', e)"
}
for rep in 1 2; do
    one cfg3 20 plan_by_work=0 proofs$rep
    one cfg3 20 plan_by_work=2 default$rep
done
one cfg3 320 plan_by_work=0 proofs
one cfg3 320 plan_by_work=2 default
one cfg4 20 plan_by_work=2 default
one cfg2 20 plan_by_work=2 default
# counters: launch 1 alone (one chain of 5120 proofs, one stream), three passes
cd /tmp && export TMPDIR=/tmp
ALONE="python $REPO/bench.py --no-cpu-baseline --no-extra --direct --streams 1 --batch 5120 --steps 6 --warmup 2 --opt horner_lanes=1"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_LDS"; do
    i=$((i+1)); rm -rf /tmp/s1pm_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/s1pm_$i -o t --output-format csv -- $ALONE > /tmp/s1pm_$i.log 2>&1 || echo "pmc set '$set' failed" >> $OUT/pmc_errors.txt
done
python3 - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/s1pm_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("$OUT/stage1_alone_counters_per_launch.json", "w"), indent=1)
for k, cs in sorted(out.items()):
    if "stage1" in k: print(k, {c: round(x) for c, x in cs.items()})
PY
rm -rf /tmp/s1ks; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/s1ks -o t --output-format csv -- $ALONE > /tmp/s1ks.log 2>&1
cp $(find /tmp/s1ks -name "*kernel_stats.csv" | head -1) $OUT/wide_chain_alone_kernel_stats.csv
grep -E "stage1" $OUT/wide_chain_alone_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-200
cat $OUT/pmc_errors.txt 2>/dev/null
# config 5's counters with the heavy pass in the chain
cd $REPO && PMC_CFGS=cfg5 ONLY_PMC=1 bash tools/collect_profiles.sh r05 > $OUT/collect_cfg5.log 2>&1; tail -2 $OUT/collect_cfg5.log | cut -c1-400

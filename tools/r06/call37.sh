#!/bin/bash
# round 6, call 37: final evidence for the narrow chain: rocprofv3 kernel stats of the call-shape regimes, the queue's timeline under tickets, a soak of
# the combining queue (9 rounds x 12 regimes, every result compared with the oracle's), VALU work of a one-proof chain's kernels
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call37
mkdir -p $OUT
cd $REPO
bash tools/combine_soak.sh 9 > $OUT/soak.log 2>&1; tail -2 $OUT/soak.log; cp gpurun_out/soak/summary.txt $OUT/soak_summary.txt; [ -f gpurun_out/soak/failures.txt ] && cp gpurun_out/soak/failures.txt $OUT/soak_failures.txt
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for mode in "threads 1" "threads 64" "tickets 16 128"; do
  name=$(echo $mode | tr ' ' '_')
  rm -rf /tmp/pf_$name
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -o t --output-format csv -- /tmp/combine_rate $INP 1.0 $mode > /tmp/pf_$name.log 2>&1
  cp $(find /tmp/pf_$name -name "*kernel_stats.csv" | head -1) $OUT/narrow_${name}_kernel_stats.csv
  grep -E '^\{' /tmp/pf_$name.log > $OUT/narrow_${name}_under_rocprof.json
done
BP_TRACE=$OUT/trace_tickets.jsonl /tmp/combine_rate $INP 1.0 tickets 16 128 > /dev/null 2>&1
python $REPO/tools/combine_timeline.py $OUT/trace_tickets.jsonl > $OUT/timeline_tickets_16x128.txt 2>&1; rm -f $OUT/trace_tickets.jsonl
BP_TRACE=$OUT/trace_t64.jsonl /tmp/combine_rate $INP 1.0 threads 64 > /dev/null 2>&1
python $REPO/tools/combine_timeline.py $OUT/trace_t64.jsonl > $OUT/timeline_threads_64.txt 2>&1; rm -f $OUT/trace_t64.jsonl
BP_TRACE=$OUT/trace_t1.jsonl /tmp/combine_rate $INP 0.5 threads 1 > /dev/null 2>&1
python $REPO/tools/combine_timeline.py $OUT/trace_t1.jsonl > $OUT/timeline_threads_1.txt 2>&1; rm -f $OUT/trace_t1.jsonl
# VALU work of the one-proof chain (separate counter pass, kernel trace only)
rm -rf /tmp/pm_valu
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d /tmp/pm_valu -o t --output-format csv -- /tmp/combine_rate $INP 0.3 threads 1 > /tmp/pm_valu.log 2>&1
python - <<'PY' > $OUT/narrow_one_proof_valu.txt 2>&1
import csv, glob, collections
f = glob.glob("/tmp/pm_valu/**/*counter_collection.csv", recursive=True)[0]
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    by[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("one-proof chain (threads 1): per launch, median over the run")
tot = 0
for k, d in sorted(by.items()):
    v = sorted(d.get("SQ_INSTS_VALU", [0])); w = sorted(d.get("SQ_WAVES", [0]))
    print("  %-28s launches %5d  SQ_INSTS_VALU %10.0f  SQ_WAVES %6.0f" % (k[:28], len(v), v[len(v) // 2], w[len(w) // 2]))
    if "stage" in k or "finish" in k: tot += v[len(v) // 2]
print("  wavefront-instructions per one-proof verification (its four launches): %.0f" % tot)
PY
cat $OUT/narrow_one_proof_valu.txt; cat $OUT/timeline_tickets_16x128.txt | head -20

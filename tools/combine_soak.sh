#!/bin/bash
# GPU box: many short runs of the combining queue's call shapes -- the regimes in which buffers fill, reopen and change width most often -- to
# catch what only shows once in a few dozen runs (round 5: a reopened buffer sealed on its predecessor's word, 3 of 23 runs).  Every run has the
# client's watchdog; a run without a result line, with mismatches, errors or a non-zero exit code is listed.  Writes gpurun_out/soak/*.
#   bash tools/combine_soak.sh [rounds]
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/soak
mkdir -p $OUT
ROUNDS=${1:-4}
export GPU_MAX_HW_QUEUES=16 BP_LANES=8 BP_W=14
g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,$PWD/bulletproofs_amd/csrc -o /tmp/combine_rate || exit 1
INP=bench_data/combine_rate_inputs.bin
python3 tools/make_msm_inputs.py /tmp/soak_msm_inputs.bin > /dev/null 2>&1 && export BP_MSM_INPUTS=/tmp/soak_msm_inputs.bin
bad=0; n=0
for r in $(seq 1 $ROUNDS); do
  for spec in "threads 256 1" "tickets 16 512" "threads 64 3" "tickets 4 512" "tickets 16 128|combine_inflight=2,combine_wait_us=300" "threads 256 1|combine_busy_chains=1" \
              "tickets 16 512|combine_inflight=3" "threads 128 2|combine_max_open=1" "big 4 1500" "tickets 32 64|combine_mapped_out=0" \
              "msm 48 1" "msm 24 3|combine_msm_bytes=4194304"; do
    mode=${spec%%|*}; opts=""; [ "$spec" != "$mode" ] && opts=${spec#*|}
    n=$((n+1))
    w=$BP_W; [ "${mode%% *}" = "msm" ] && w=10      # (the MSM rows need BulletproofGens(2048, 1): small windows keep the table build short)
    BP_W=$w BP_OPTS=$opts timeout 60 /tmp/combine_rate $INP 0.6 $mode > $OUT/run.json 2> $OUT/run.err; rc=$?
    line=$(grep '^{' $OUT/run.json | tail -1)
    ok=$(python3 -c "
import json,sys
try:
    d=json.loads(sys.argv[1]); print('ok' if d['mismatches']==0 and d['errors']==0 and d['rate_per_s']>0 else 'BAD')
except Exception: print('NOLINE')" "$line")
    if [ "$rc" != "0" ] || [ "$ok" != "ok" ]; then
      bad=$((bad+1)); echo "round $r [$mode] opts=[$opts]: rc=$rc $ok" | tee -a $OUT/failures.txt; cat $OUT/run.err >> $OUT/failures.txt
    fi
  done
done
echo "combine_soak: $n runs, $bad failed" | tee $OUT/summary.txt

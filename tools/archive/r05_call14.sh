#!/bin/bash
# Round 5: the narrow chain's scalar role with its U coefficient recodings on U lanes (option coop_defer_emit) against the leader-serial form:
# parity tests with the option on, call latency of one / 16 / 64 blocking callers, the kernel's duration under the trace.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05h
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,$PWD/bulletproofs_amd/csrc -o /tmp/combine_rate || exit 1
INP=bench_data/combine_rate_inputs.bin
run() {
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env BP_LANES=8 BP_W=16 "${envs[@]}" timeout 60 /tmp/combine_rate $INP 1.5 "$@" > $OUT/$name.json 2>> $OUT/log.txt
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
for rep in 1 2 3; do
  for de in 0 1; do
    run t1_defer${de}_$rep BP_OPTS=coop_defer_emit=$de -- threads 1
    run t16_defer${de}_$rep BP_OPTS=coop_defer_emit=$de -- threads 16
    run t64_defer${de}_$rep BP_OPTS=coop_defer_emit=$de -- threads 64
  done
done
run tk_defer0 BP_OPTS=coop_defer_emit=0 -- tickets 16 128
run tk_defer1 BP_OPTS=coop_defer_emit=1 -- tickets 16 128
for de in 0 1; do
  (cd /tmp && export TMPDIR=/tmp && BP_LANES=8 BP_W=16 BP_OPTS=coop_defer_emit=$de timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pf_de$de -o t --output-format csv -- /tmp/combine_rate $OLDPWD/$INP 0.5 threads 1 > /dev/null 2>&1; grep -E "stage1_coop|stage3|stage4|finish8" $(find /tmp/pf_de$de -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4,6 | sed "s/(bp::[^\"]*\"/\"/" | cut -c1-110 | sed "s/^/defer=$de  /")
done
# parity with the option on for every context of the process (BPGPU_COOP_DEFER_EMIT): the suites that run narrow chains
BPGPU_COOP_DEFER_EMIT=1 timeout 600 python -m pytest tests/test_gpu_transcript_coop.py tests/test_gpu_combine.py tests/test_gpu_rangeproof.py tests/test_gpu_transcript_stop.py tests/test_gpu_rlc.py -q 2>&1 | tail -3

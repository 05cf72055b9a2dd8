#!/bin/bash
# round 6, call 44: four table levels per point (16-window chain): parity, A/B of narrow_hi4_max, kernel timeline
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call44
mkdir -p $OUT
cd $REPO
timeout 2000 python -m pytest tests/test_gpu_narrow_chain.py tests/test_gpu_transcript_coop.py tests/test_gpu_rangeproof.py tests/test_gpu_transcript_stop.py tests/test_gpu_combine.py tests/test_gpu_pool.py tests/test_gpu_coalesce_shapes.py tests/test_gpu_reference_api.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
run() { # env-string mode...
  local e=$1; shift
  env $e /tmp/combine_rate $INP 1.0 "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', '$*', 'rate', d['rate_per_s'], 'lat', d['lat_ms'], 'per chain', d['proofs_per_chain'], 'mismatches', d['mismatches'], 'errors', d['errors'])" >> $OUT/narrow_ab.txt
}
for rep in 1 2 3; do
  for e in "BPGPU_NARROW_HI4_MAX=0" "BPGPU_NARROW_HI4_MAX=2" "BPGPU_NARROW_HI4_MAX=8" "BPGPU_NARROW_HI4_MAX=32"; do
    run "$e" threads 1
    run "$e" threads 16
    run "$e" threads 64
    run "$e" threads 256
    run "$e" tickets 16 128
  done
done
cat $OUT/narrow_ab.txt
timeline() { # label
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- /tmp/combine_rate $INP 0.5 threads 1 > $OUT/trace.log 2>&1
  f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
  python - "$f" "$1" >> $OUT/single_call_kernel_timeline.txt <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
rows = rows[len(rows) // 2:]   # steady state
by = collections.defaultdict(list)
for s, e, k in rows: by[k].append((e - s) / 1e3)
print("%s: one blocking caller, one proof per call; median kernel durations over the second half of the run" % sys.argv[2])
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort(); print("  %-28s n %5d  p50 %7.1f us" % (k[:28], len(v), v[len(v) // 2]))
st = [s for s, e, k in rows if "stage1" in k]
per = sorted((st[i + 1] - st[i]) / 1e3 for i in range(len(st) - 1))
print("  call period p50 %.1f us" % per[len(per) // 2])
PY
  rm -rf $OUT/trace
}
BPGPU_NARROW_HI4_MAX=0 timeline "narrow_hi4_max=0"
BPGPU_NARROW_HI4_MAX=32 timeline "narrow_hi4_max=32"
cat $OUT/single_call_kernel_timeline.txt

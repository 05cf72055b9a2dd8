#!/bin/bash
# round 6, call 15: where the latency maxima of the call-shape rows sit (outlier report of tools/combine_rate.cpp), the option-flip test, the table curve
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call14
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_pool.py -x -q -m gpu -k "option or coalesced or one_call" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -o /tmp/cr -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,$REPO/bulletproofs_amd/csrc 2>&1 | tail -3
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for mode in "threads 1" "threads 64" "threads 256" "tickets 16 128" "big 2 4096"; do
  echo "== $mode" >> $OUT/outliers.txt
  BP_TRACE=/tmp/trace_$(echo $mode | tr ' ' '_').txt /tmp/cr bench_data/combine_rate_inputs.bin 2.0 $mode >> $OUT/outliers.txt 2>&1
done
cut -c1-900 $OUT/outliers.txt
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
j = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
e = j["extra"]
print("headline", j["value"]); print(json.dumps(e.get("table_curve"))[:1500])
for k, v in e["drop_in_call_shape"].items():
    if isinstance(v, dict): print(k, v.get("verifications_per_s", v.get("msms_per_s")), v.get("latency_ms"), v.get("steady_latency_ms"), v.get("outliers_above_5x_p99"), [x for kk, x in v.items() if kk.startswith("of_them")])
print("cfg5", {k: v for k, v in e["cfg5_shape"].items() if isinstance(v, (int, float))}, e["cfg5_shape"]["roofline"].get("valu"), e["cfg5_shape"]["roofline"].get("hbm_counter"))
PY

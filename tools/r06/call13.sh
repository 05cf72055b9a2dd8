#!/bin/bash
# round 6, call 13: after the option pruning -- the pool's GPU suites, then the service thread's scheduling class A/B on the call-shape rows
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call13
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_pool.py tests/test_gpu_pool_msm.py tests/test_gpu_combine.py tests/test_gpu_coalesce_shapes.py tests/test_gpu_concurrency.py tests/test_gpu_mixed_shapes.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for sched in default rr default rr; do
  if [ $sched = rr ]; then export BPGPU_SERVICE_SCHED=rr; else unset BPGPU_SERVICE_SCHED; fi
  python $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_$sched.json 2> $OUT/bench_$sched.err
  python - <<PY >> $OUT/sched_ab.txt
import json
j = json.loads([l for l in open("$OUT/bench_$sched.json") if l.startswith("{")][-1])
d = j["extra"]["drop_in_call_shape"]
print("$sched", "headline", j["value"], " ".join("%s %s p50 %s p99 %s max %s |" % (k, v.get("verifications_per_s", v.get("msms_per_s")), v["latency_ms"]["p50"], v["latency_ms"]["p99"], v["latency_ms"]["max"]) for k, v in d.items() if isinstance(v, dict) and "latency_ms" in v))
PY
done
cat $OUT/sched_ab.txt

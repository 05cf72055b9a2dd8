#!/bin/bash
# round 6, call 55: the exponent launch at three wavefronts per SIMD for aggregated shapes: steady and burst forms A/B (BPGPU_EXP_W3_MIN_NM), parity of cfg3 / cfg4
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call55
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
one() { # env args
  local e=$1; shift
  env $e python $REPO/bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$e', '$*', j['value'], j['ms_per_step'])" >> $OUT/exp_w3_ab.txt
}
for rep in 1 2 3; do
  for e in BPGPU_EXP_W3_MIN_NM=1000000 BPGPU_EXP_W3_MIN_NM=1024; do
    one $e --config cfg3 --steps 640 --warmup 64
    one $e --config cfg3 --steps 20 --warmup 5
    one $e --config cfg4 --steps 320 --warmup 32
    one $e --config cfg4 --steps 20 --warmup 5
  done
done
cat $OUT/exp_w3_ab.txt
cd $REPO; timeout 1500 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_rangeproof.py tests/test_gpu_coalesce_shapes.py -x -q -m gpu 2>&1 | tail -2

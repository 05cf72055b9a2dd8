#!/bin/bash
# round 6, call 52: wide chains with the generator exponents on the second stream first (exp_early): same-box A/B of the bench forms + parity
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call52
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
one() { # env args
  local e=$1; shift
  env $e python $REPO/bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$e', '$*', j['value'], j['ms_per_step'])" >> $OUT/exp_early_ab.txt
}
for rep in 1 2 3; do
  for e in BPGPU_EXP_EARLY=0 BPGPU_EXP_EARLY=1; do
    one $e
    one $e --steps 20 --warmup 5
    one $e --config cfg3 --steps 20 --warmup 5
    one $e --config cfg3 --steps 640 --warmup 64
    one $e --config cfg4 --steps 20 --warmup 5
  done
done
cat $OUT/exp_early_ab.txt
cd $REPO; timeout 1500 python -m pytest tests/test_gpu_rangeproof.py tests/test_gpu_bench_config.py tests/test_gpu_pool.py tests/test_gpu_concurrency.py -x -q -m gpu 2>&1 | tail -2

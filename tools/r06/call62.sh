#!/bin/bash
# round 6, call 62: a small leading chain in a burst (experiment BPGPU_FIRST_CHAIN): the 20-step forms of cfg2 / cfg3 / cfg4
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call62
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
one() { # env args
  local e=$1; shift
  env $e python $REPO/bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$e', '$*', j['value'], j['ms_per_step'])" >> $OUT/first_chain_ab.txt
}
for rep in 1 2 3; do
  for n in 0 1024 2048 4096; do
    one BPGPU_FIRST_CHAIN=$n --steps 20 --warmup 5
    one BPGPU_FIRST_CHAIN=$n --steps 20 --warmup 5
  done
  for n in 0 256 512 1024; do
    one BPGPU_FIRST_CHAIN=$n --config cfg3 --steps 20 --warmup 5
    one BPGPU_FIRST_CHAIN=$n --config cfg4 --steps 20 --warmup 5
  done
done
cat $OUT/first_chain_ab.txt

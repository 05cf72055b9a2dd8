#!/bin/bash
# round 6, call 50: launch 1's decode role with the short-register decode (ristretto_decompress_lp; spills of k_rp_stage1<true> 117 -> 81): same-box A/B of the bench forms
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call50
mkdir -p $OUT
LIB=$REPO/bulletproofs_amd/csrc
mkdir -p /tmp/var_lp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -DBP_POINTS_LP -c -o /tmp/var_lp/k_rp1.o $LIB/k_rp1.hip || exit 1
objs=$(ls $LIB/build/*.o | grep -v k_rp1.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/var_lp/libbpgpu.so $objs /tmp/var_lp/k_rp1.o || exit 1
cp $LIB/libbpgpu.so /tmp/lib_base.so
cd /tmp && export TMPDIR=/tmp
one() { # label args
  local label=$1; shift
  python $REPO/bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$label', '$*', j['value'], j['ms_per_step'])" >> $OUT/points_lp_ab.txt
}
for rep in 1 2 3; do
  for v in base lp; do
    if [ $v = lp ]; then cp /tmp/var_lp/libbpgpu.so $LIB/libbpgpu.so; else cp /tmp/lib_base.so $LIB/libbpgpu.so; fi
    one $v
    one $v --steps 20 --warmup 5
    one $v --config cfg3 --steps 640 --warmup 64
  done
done
cp /tmp/lib_base.so $LIB/libbpgpu.so
cat $OUT/points_lp_ab.txt
cd $REPO; cp /tmp/var_lp/libbpgpu.so $LIB/libbpgpu.so; timeout 900 python -m pytest tests/test_gpu_rangeproof.py tests/test_gpu_bench_config.py -x -q -m gpu 2>&1 | tail -2; cp /tmp/lib_base.so $LIB/libbpgpu.so

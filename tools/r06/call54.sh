#!/bin/bash
# round 6, call 54: the generator-exponent launch of wide chains at three wavefronts per SIMD (the four-index form without spills, the paired form with 27
# spilled registers) against the default (paired form, two wavefronts): same-box A/B of the bench forms
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call54
mkdir -p $OUT
LIB=$REPO/bulletproofs_amd/csrc
mkdir -p /tmp/var_w3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -DBP_EXP_WAVES=3 -c -o /tmp/var_w3/k_rp34.o $LIB/k_rp34.hip || exit 1
objs=$(ls $LIB/build/*.o | grep -v k_rp34.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/var_w3/libbpgpu.so $objs /tmp/var_w3/k_rp34.o || exit 1
cp $LIB/libbpgpu.so /tmp/lib_base.so
cd /tmp && export TMPDIR=/tmp
one() { # label args
  local label=$1; shift
  python $REPO/bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$label', '$*', j['value'], j['ms_per_step'])" >> $OUT/exp_waves_ab.txt
}
for rep in 1 2 3; do
  for v in "base none" "w3 exponent_pairs=1" "w3 exponent_pairs=0"; do
    set -- $v
    if [ $1 = w3 ]; then cp /tmp/var_w3/libbpgpu.so $LIB/libbpgpu.so; else cp /tmp/lib_base.so $LIB/libbpgpu.so; fi
    opt=""; [ $2 != none ] && opt="--opt $2"
    one "$1/$2" $opt
    one "$1/$2" --steps 20 --warmup 5 $opt
    one "$1/$2" --steps 20 --warmup 5 $opt
    one "$1/$2" --config cfg3 --steps 20 --warmup 5 $opt
    one "$1/$2" --config cfg4 --steps 20 --warmup 5 $opt
  done
done
cp /tmp/lib_base.so $LIB/libbpgpu.so
cat $OUT/exp_waves_ab.txt

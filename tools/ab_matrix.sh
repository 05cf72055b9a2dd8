#!/bin/bash
# GPU box: interleaved A/B over (library build) x (option string) on the two bench forms.  LIBS="a b" OPTS="none k=v ..." tools/ab_matrix.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export GPU_MAX_HW_QUEUES=16
mkdir -p $REPO/gpurun_out/r04
OUT=$REPO/gpurun_out/r04/ab_matrix_${TAG:-generic}.txt
cp bulletproofs_amd/csrc/libbpgpu.so /tmp/keep.so
B="python $REPO/bench.py --no-cpu-baseline --no-extra"
for r in $(seq ${ROUNDS:-3}); do
  for v in $LIBS; do
    cp ab/$v.so bulletproofs_amd/csrc/libbpgpu.so
    for o in $OPTS; do
      for args in ${FORMS:-"--steps=20,--warmup=5" "-"}; do
        A=$(echo $args | tr ',' ' '); [ "$args" = "-" ] && A=""
        if [ "$o" = "none" ]; then OPT=""; else OPT="--opt $o"; fi
        $B $A $OPT 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', '$o', '[$A]', round(d['value']), {k: round(x) for k, x in ((d.get('roofline') or {}).get('kernels_us') or {}).items()})" >> $OUT
      done
    done
  done
done
cp /tmp/keep.so bulletproofs_amd/csrc/libbpgpu.so
cat $OUT

#!/bin/bash
# GPU box: interleaved A/B of library options on the two bench forms.  tools/r04_ab_opts.sh "optA" "optB" ...   (opt = key=val[,key=val] or "none")
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export GPU_MAX_HW_QUEUES=16
mkdir -p $REPO/gpurun_out/r04
OUT=$REPO/gpurun_out/r04/ab_opts_${TAG:-generic}.txt
B="python $REPO/bench.py --no-cpu-baseline --no-extra"
for r in $(seq ${ROUNDS:-3}); do
  for o in "$@"; do
    for args in "--steps 20 --warmup 5" ""; do
      if [ "$o" = "none" ]; then OPT=""; else OPT="--opt $o"; fi
      $B $args $OPT 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$o', '[$args]', round(d['value']), {k: round(x) for k, x in ((d.get('roofline') or {}).get('kernels_us') or {}).items()})" >> $OUT
    done
  done
done
cat $OUT

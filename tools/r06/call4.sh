#!/bin/bash
# counters of the config-5 chain by (kernel, grid): VALU work, HBM fetch / write
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OPT=${OPT:-bucket_chain=0}
for c in SQ_INSTS_VALU FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_$c -o t --output-format csv -- python $REPO/bench.py --cfg5-only 1 --opt $OPT > /tmp/pm_$c.log 2>&1
  echo "== $c ($OPT)" >> $OUT/pmc_by_grid.txt
  python $REPO/tools/r06/pmc_by_grid.py /tmp/pm_$c $c >> $OUT/pmc_by_grid.txt
done
cat $OUT/pmc_by_grid.txt

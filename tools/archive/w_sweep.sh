#!/bin/bash
# Window-width sweep on the aggregated shapes (BASELINE configs 3 and 4): throughput through the pool at each W, and -- separate
# rocprofv3 --pmc passes, kernel-trace only -- the L2 hit / miss counts and HBM read requests of the table-walk launch.
# Small W keeps a (generator, window) sub-table in the XCD's L2 for more additions; large W is fewer additions, all HBM gathers.
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/wsweep_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-extra --table-bytes 200000000000"
for spec in "cfg3 10 12 13 14 15 16" "cfg4 10 12 13 14 15"; do
  set -- $spec; cfg=$1; shift
  for W in "$@"; do
    $B --config $cfg --window-bits $W --steps 320 --warmup 32 --no-events 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg W=$W rate', round(d['value']), 'table_GB', round(d['config']['fixed_table_bytes']/1e9,1))" | tee -a $OUT/rates.txt
    for ctr in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_WAVES"; do
      tag=$(echo $ctr | tr ' ' '_')
      rm -rf /tmp/ws_pmc
      rocprofv3 --kernel-trace --pmc $ctr -d /tmp/ws_pmc -o t --output-format csv -- $B --direct --config $cfg --window-bits $W --steps 4 --warmup 1 --streams 1 > /tmp/ws_pmc.log 2>&1
      python - <<PY | tee -a $OUT/pmc.txt
import csv, glob, collections
f = glob.glob("/tmp/ws_pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
if f:
    for r in csv.DictReader(open(f[0])):
        if "rp_stage4" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("$cfg W=$W rp_stage4", {k: round(v[1] / v[0], 1) for k, v in acc.items()})
PY
    done
  done
done

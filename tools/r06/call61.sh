#!/bin/bash
# round 6, call 61: rocprofv3 kernel stats of the two driver forms on the final tree
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call61
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-extra"
stats() {
  local name=$1; shift
  rm -rf /tmp/pf_$name
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -o t --output-format csv -- "$@" > /tmp/pf_$name.log 2>&1
  cp $(ls -S $(find /tmp/pf_$name -name "*kernel_stats.csv") | head -1) $OUT/${name}_kernel_stats.csv
  grep -E '^\{' /tmp/pf_$name.log > $OUT/${name}_under_rocprof.json
}
stats bench_default $B
stats bench_steps20 $B --steps 20 --warmup 5
stats bench_cfg3 $B --config cfg3 --steps 640 --warmup 64
stats bench_cfg4 $B --config cfg4 --steps 320 --warmup 32
for n in bench_default bench_steps20 bench_cfg3 bench_cfg4; do python -c "
import json, csv
j = json.loads(open('$OUT/${n}_under_rocprof.json').read().strip().splitlines()[-1])
print('$n', j['value'], j['ms_per_step'], 'roofline avg_launch_us', j['roofline'].get('avg_launch_us'), j['roofline'].get('frac'))
for r in csv.DictReader(open('$OUT/${n}_kernel_stats.csv')):
    if 'stage4' in r['Name'] or 'exponents' in r['Name']: print('   ', r['Name'].split('(')[0], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')"; done

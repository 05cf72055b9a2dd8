#!/bin/bash
# round 6, call 43: the driver's two bench forms on the final tree (not under the profiler), kernel stats of the default form
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call43
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
rm -rf /tmp/pf_default
rocprofv3 --kernel-trace --stats -d /tmp/pf_default -o t --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-extra > /tmp/pf_default.log 2>&1
cp $(find /tmp/pf_default -name "*kernel_stats.csv" | head -1) $OUT/bench_default_kernel_stats.csv
grep -E '^\{' /tmp/pf_default.log > $OUT/bench_default_under_rocprof.json
for f in bench_default bench_steps20; do python -c "
import json
j = json.loads([l for l in open('$OUT/$f.json') if l.startswith('{')][-1])
print('$f', j['value'], j['ms_per_step'], j['roofline'], {k: (v.get('verifications_per_s') or v.get('msms_per_s'), v['latency_ms']['p50'], v['latency_ms']['p99'], v['latency_ms']['max']) for k, v in j['extra']['drop_in_call_shape'].items() if isinstance(v, dict) and 'latency_ms' in v})
print({k: v.get('verifications_per_s', v.get('msms_per_s')) for k, v in j['extra'].items() if isinstance(v, dict) and k in ('cfg3', 'cfg4', 'cfg5_shape', 'rlc', 'rlc_batch4096')}, j['extra'].get('msm_small_single_call'))"; done

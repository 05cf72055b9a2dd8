#!/bin/bash
# round 6, call 48: soak of the combining queue on the final tree (9 rounds x 12 regimes, every result compared with the oracle's) + the driver's bench forms
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call48
mkdir -p $OUT
cd $REPO
bash tools/combine_soak.sh 9 > $OUT/soak.log 2>&1; tail -2 $OUT/soak.log; cp gpurun_out/soak/summary.txt $OUT/soak_summary.txt; [ -f gpurun_out/soak/failures.txt ] && cp gpurun_out/soak/failures.txt $OUT/soak_failures.txt
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
for f in bench_default bench_steps20; do python -c "
import json
j = json.loads([l for l in open('$OUT/$f.json') if l.startswith('{')][-1])
print('$f', j['value'], j['ms_per_step'], {k: (v.get('verifications_per_s') or v.get('msms_per_s'), v['latency_ms']['p50'], v['latency_ms']['p99'], v['latency_ms']['max']) for k, v in j['extra']['drop_in_call_shape'].items() if isinstance(v, dict) and 'latency_ms' in v})
print({k: v.get('verifications_per_s', v.get('msms_per_s')) for k, v in j['extra'].items() if isinstance(v, dict) and k in ('cfg3', 'cfg4', 'cfg5_shape', 'rlc', 'rlc_batch4096')})"; done

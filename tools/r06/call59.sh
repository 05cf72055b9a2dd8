#!/bin/bash
# round 6, call 59: final tree -- whole GPU suite, a short soak, the default bench line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call59
mkdir -p $OUT
cd $REPO
timeout 3000 python -m pytest tests -q -m gpu > $OUT/pytest_gpu_full.txt 2>&1; tail -4 $OUT/pytest_gpu_full.txt
bash tools/combine_soak.sh 4 > $OUT/soak.log 2>&1; tail -1 $OUT/soak.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
for f in bench_default bench_steps20; do python -c "
import json
j = json.loads([l for l in open('$OUT/$f.json') if l.startswith('{')][-1])
print('$f', j['value'], j['ms_per_step'], {k: (v.get('verifications_per_s') or v.get('msms_per_s'), v['latency_ms']['p50'], v['latency_ms']['p99'], v['latency_ms']['max']) for k, v in j['extra']['drop_in_call_shape'].items() if isinstance(v, dict) and 'latency_ms' in v})
print({k: v.get('verifications_per_s', v.get('msms_per_s')) for k, v in j['extra'].items() if isinstance(v, dict) and k in ('cfg3', 'cfg4', 'cfg5_shape', 'rlc', 'rlc_batch4096')})"; done

#!/bin/bash
# round 6, call 41: whole GPU suite + smoke + the default bench line on the tree with the narrow small-MSM form
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call41
mkdir -p $OUT
cd $REPO
timeout 3000 python -m pytest tests -q -m gpu > $OUT/pytest_gpu_full.txt 2>&1; tail -4 $OUT/pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json
j = json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1])
print('default:', j['value'], j['ms_per_step'], {k: (v.get('verifications_per_s') or v.get('msms_per_s'), v['latency_ms']['p50'], v['latency_ms']['p99'], v['latency_ms']['max']) for k, v in j['extra']['drop_in_call_shape'].items() if isinstance(v, dict) and 'latency_ms' in v})
print(json.dumps(j['extra'].get('msm_small_single_call')))
print({k: v.get('verifications_per_s', v.get('msms_per_s')) for k, v in j['extra'].items() if isinstance(v, dict) and k in ('cfg3', 'cfg4', 'cfg5_shape', 'rlc', 'rlc_batch4096')})"

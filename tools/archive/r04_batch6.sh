#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_rlc.py -x -q > gpurun_out/r04/test_batch6.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r04/test_batch6.txt
for args in "--steps 20 --warmup 5" ""; do
timeout 900 python bench.py --no-cpu-baseline $args > gpurun_out/r04/bench_b6.json 2> gpurun_out/r04/bench_b6.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_b6.json').read().strip().splitlines()[-1])
ex=d.get('extra',{})
print(d['steps'], d['value'], {k:(v.get('verifications_per_s') or v.get('proofs_per_s') or v.get('msms_per_s') or v.get('error')) for k,v in ex.items() if isinstance(v,dict)})
print(json.dumps(ex.get('rlc'))[:300]); print(json.dumps(ex.get('rlc_batch4096'))[:600])
PY
done

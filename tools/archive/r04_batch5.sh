#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests/test_gpu_rlc.py tests/test_gpu_pool.py -x -q > gpurun_out/r04/test_batch5.txt 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r04/test_batch5.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_steps20_b5.json 2> gpurun_out/r04/bench_steps20_b5.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_steps20_b5.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['bound'], d['roofline']['frac'])
for k,v in d.get('extra',{}).items():
    print(k, json.dumps(v)[:400])
print(d.get('cpu_baseline'))
PY
tail -3 gpurun_out/r04/bench_steps20_b5.err

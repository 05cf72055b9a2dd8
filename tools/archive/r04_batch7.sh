#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests/test_gpu_mixed_shapes.py tests/test_gpu_rlc.py tests/test_gpu_prover_ct.py -x -q > gpurun_out/r04/test_batch7.txt 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r04/test_batch7.txt
timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04/bench_b7.json 2> gpurun_out/r04/bench_b7.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_b7.json').read().strip().splitlines()[-1])
ex=d.get('extra',{})
print(d['steps'], d['value'], {k:(v.get('verifications_per_s') or v.get('proofs_per_s') or v.get('msms_per_s') or v.get('error')) for k,v in ex.items() if isinstance(v,dict)})
print(json.dumps(ex.get('mixed_shapes'))[:900]); print(json.dumps(ex.get('drop_in_call_shape'))[:1200])
PY
tail -3 gpurun_out/r04/bench_b7.err

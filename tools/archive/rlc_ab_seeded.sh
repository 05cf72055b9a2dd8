#!/bin/bash
mkdir -p gpurun_out/r04
OUT=$PWD/gpurun_out/r04/ab_rlc_seeded.txt
: > $OUT
timeout 300 python -m pytest tests/test_gpu_rlc.py tests/test_gpu_rangeproof.py -x -q 2>&1 | tail -3 >> $OUT
run() { python bench.py --rlc --batch 4096 --steps 24 --no-extra --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', 'value=%.0f' % d['value'], json.dumps(d['roofline'].get('kernels_us')))
" >> $OUT 2>&1; }
run --streams 1
run --streams 12
run --streams 12
run --streams 32 --steps 256
(cd ab/r03; run --streams 32 --steps 256)
python bench.py --rlc --batch 1024 --steps 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pool rlc batch1024 steps20 value=%.0f' % d['value'], json.dumps(d['roofline'].get('kernels_us')))" >> $OUT
python bench.py --rlc --batch 1024 --steps 256 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pool rlc batch1024 steps256 value=%.0f' % d['value'], json.dumps(d['roofline'].get('kernels_us')))" >> $OUT
cat $OUT

#!/bin/bash
cp bulletproofs_amd/csrc/libbpgpu.so /tmp/keep.so
for r in 1 2; do for v in base2 caps; do cp ab/$v.so bulletproofs_amd/csrc/libbpgpu.so
  python bench.py --cfg5-only 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v cfg5 x8', j['msms_per_s'], j['ms_single_msm'])"
  python bench.py --rlc --no-extra --no-cpu-baseline --steps 1280 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v rlc', j['value'])"
  python bench.py --rlc --batch 4096 --streams 32 --steps 256 --warmup 32 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v rlc4096', j['value'])"
done; done
cp /tmp/keep.so bulletproofs_amd/csrc/libbpgpu.so

#!/bin/bash
# same-box A/B of two library builds on the aggregated shapes (cfg3: m = 16 batch 256; cfg4 shape: m = 32 batch 512)
cp bulletproofs_amd/csrc/libbpgpu.so /tmp/keep.so
for r in 1 2; do for v in "$@"; do cp ab/$v.so bulletproofs_amd/csrc/libbpgpu.so
  python bench.py --config cfg3 --steps 640 --warmup 64 --streams 64 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v cfg3', j['value'])"
  python bench.py --config cfg4 --steps 256 --warmup 32 --streams 32 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v cfg4', j['value'])"
done; done
cp /tmp/keep.so bulletproofs_amd/csrc/libbpgpu.so

#!/bin/bash
# same-box A/B of two library builds at the driver's short run (--steps 20 --warmup 5) and at the default run
cp bulletproofs_amd/csrc/libbpgpu.so /tmp/keep.so
for r in 1 2 3; do for v in "$@"; do cp ab/$v.so bulletproofs_amd/csrc/libbpgpu.so
  python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v steps20', j['value'])"
  python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v default', j['value'])"
done; done
cp /tmp/keep.so bulletproofs_amd/csrc/libbpgpu.so

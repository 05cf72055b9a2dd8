#!/bin/bash
# same-box A/B of environment settings: each argument is a "VAR=val VAR2=val" string ("-" for none)
for r in 1 2; do for v in "$@"; do
  e=""; [ "$v" != "-" ] && e="$v"
  env $e python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('[$v] steps20', j['value'])"
  env $e python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('[$v] default', j['value'])"
done; done

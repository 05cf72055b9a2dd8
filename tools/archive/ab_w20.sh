#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-extra --no-events "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', '->', round(d['value']), 'table GB', round(d['config']['fixed_table_bytes']/1e9,1))"; }
for r in 1 2 3; do
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --window-bits 20 --table-bytes 130000000000
done
run
run --window-bits 20 --table-bytes 130000000000

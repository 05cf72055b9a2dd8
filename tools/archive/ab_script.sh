#!/bin/bash
# A/B of the scripted transcript (csrc/rp_script.h) against the byte-wise replay: launch 1 alone (one stream), the driver form, the default run
run() { python bench.py --no-cpu-baseline --no-extra "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', '->', round(d['value']), (d.get('roofline') or {}).get('kernels_us'))"; }
for r in 1 2; do
run --direct --streams 1 --steps 64 --warmup 8 --events-all
run --direct --streams 1 --steps 64 --warmup 8 --events-all --no-script
run --steps 20 --warmup 5 --no-events
run --steps 20 --warmup 5 --no-events --no-script
done
run --no-events
run --no-events --no-script

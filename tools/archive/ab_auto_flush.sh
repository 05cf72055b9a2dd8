run() { python bench.py --no-cpu-baseline --no-extra --no-events "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']))"; }
for r in 1 2 3; do
for o in auto_flush_proofs=0 auto_flush_proofs=10240 auto_flush_proofs=8192 auto_flush_proofs=6144; do
run --steps 20 --warmup 5 --opt $o
done
done
for o in auto_flush_proofs=0 auto_flush_proofs=10240 auto_flush_proofs=6144; do run --opt $o; done

#!/bin/bash
# Round 5, the round's record on the final build: whole GPU suite, smoke, the driver's bench form and the default form, rocprofv3 kernel stats of
# the driver form and of config 5.  Writes gpurun_out/r05f/*.
set -u
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/r05f
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $OUT/pytest_gpu_full.txt
tail -3 $OUT/pytest_gpu_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench_steps20.json') if l.startswith('{')][-1]); print('steps20: %.0f /s' % d['value']); e=d['extra']
for k in ('small_table','cfg3','cfg4','rlc','rlc_batch4096','prover'): print(k, {kk:vv for kk,vv in e.get(k,{}).items() if kk in ('verifications_per_s','proofs_per_s','error','fixed_table_bytes')})
print('cfg5', {kk:vv for kk,vv in e.get('cfg5_shape',{}).items() if kk in ('msms_per_s','ms_single_msm','ms_per_batch_one_stream','error')})
print('mixed', e.get('mixed_shapes'))
for k,v in e.get('drop_in_call_shape',{}).items():
    if isinstance(v,dict): print(k, {kk:vv for kk,vv in v.items() if kk in ('verifications_per_s','msms_per_s','proofs_per_chain','msms_per_chain','error')}, v.get('latency_ms'))
print('cpu', d.get('cpu_baseline',{}).get('value'))
"
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-extra"
rm -rf /tmp/pf_s20; rocprofv3 --kernel-trace --stats -d /tmp/pf_s20 -o t --output-format csv -- $B --steps 20 --warmup 5 > /tmp/pf_s20.log 2>&1
cp $(find /tmp/pf_s20 -name "*kernel_stats.csv" | head -1) $OUT/bench_steps20_kernel_stats.csv; grep -E '^\{' /tmp/pf_s20.log > $OUT/bench_steps20_under_rocprof.json
rm -rf /tmp/pf_c5; rocprofv3 --kernel-trace --stats -d /tmp/pf_c5 -o t --output-format csv -- python $REPO/bench.py --cfg5-only 16 > /tmp/pf_c5.log 2>&1
cp $(find /tmp/pf_c5 -name "*kernel_stats.csv" | head -1) $OUT/bench_cfg5_kernel_stats.csv; grep -E '^\{' /tmp/pf_c5.log > $OUT/bench_cfg5_under_rocprof.json
head -8 $OUT/bench_steps20_kernel_stats.csv | cut -c1-120
cd $REPO
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1]); print('default: %.0f /s' % d['value']); e=d['extra']
for k in ('small_table','cfg3','cfg4','rlc','rlc_batch4096','prover'): print(k, {kk:vv for kk,vv in e.get(k,{}).items() if kk in ('verifications_per_s','proofs_per_s','error','rlc_verifications_per_s')})
print('cfg5', {kk:vv for kk,vv in e.get('cfg5_shape',{}).items() if kk in ('msms_per_s','ms_single_msm','error')})
print('mixed', e.get('mixed_shapes'))
"

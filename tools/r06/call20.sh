#!/bin/bash
# round 6, call 20: the generator-exponent role in mirrored pairs of indices (rp_expand_b8_thread): parity, then A/B on configs 3, 4, 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call20
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_rangeproof.py tests/test_gpu_bench_config.py tests/test_gpu_transcripts.py tests/test_gpu_transcript_coop.py tests/test_gpu_mixed_shapes.py tests/test_gpu_reference_api.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for cfg in cfg3 cfg4 cfg2; do
  for opt in exponent_pairs=0 exponent_pairs=1 exponent_pairs=0 exponent_pairs=1; do
    steps=640; [ $cfg = cfg4 ] && steps=256; [ $cfg = cfg2 ] && steps=3840
    python $REPO/bench.py --config $cfg --steps $steps --warmup 64 --no-extra --no-cpu-baseline --opt $opt 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('$cfg $opt:', j['value'], 'ms/step', j['ms_per_step'], {k: v for k, v in r['kernels_us'].items() if k in ('rp_stage3', 'rp_stage4')})" >> $OUT/exponent_pairs_ab.txt
  done
done
cat $OUT/exponent_pairs_ab.txt

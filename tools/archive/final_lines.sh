#!/bin/bash
# the two bench lines as the driver runs them, and rocprofv3 kernel stats of the same commands (one box, one call)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/final; mkdir -p $OUT
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
cd /tmp && export TMPDIR=/tmp
for name in default steps20; do
  args=""; [ $name = steps20 ] && args="--steps 20 --warmup 5"
  rm -rf /tmp/pf_$name
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -o t --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-extra $args > /tmp/pf_$name.log 2>&1
  cp $(find /tmp/pf_$name -name "*kernel_stats.csv" | head -1) $OUT/bench_${name}_kernel_stats.csv
  grep -E '^\{' /tmp/pf_$name.log > $OUT/bench_${name}_under_rocprof.json
done
cut -c1-160 $OUT/bench_default.json; echo; cut -c1-160 $OUT/bench_steps20.json

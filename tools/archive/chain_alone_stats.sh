cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_sat
rocprofv3 --kernel-trace --stats -d /tmp/pf_sat -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-events --direct --streams 1 --batch 16384 --steps 24 --warmup 4 --opt horner_lanes=1 > /tmp/pf_sat.log 2>&1
tail -1 /tmp/pf_sat.log | cut -c1-150
cp $(find /tmp/pf_sat -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/chain16384_alone_kernel_stats.csv
head -9 $GRAFT_REPO_ROOT/gpurun_out/chain16384_alone_kernel_stats.csv | cut -c1-60,300-420

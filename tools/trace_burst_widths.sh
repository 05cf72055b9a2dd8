cd /tmp && export TMPDIR=/tmp
for cz in 5120 10240 20480; do
rm -rf /tmp/tr_$cz
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$cz -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-events --steps 20 --warmup 5 --coalesce $cz --opt max_chain_proofs=32768 > /tmp/tr_$cz.log 2>&1
tail -1 /tmp/tr_$cz.log | cut -c1-120
echo "== coalesce $cz"
python $GRAFT_REPO_ROOT/tools/trace_burst.py /tmp/tr_$cz 1 detail
done

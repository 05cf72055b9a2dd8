#!/bin/bash
# kernel timeline of the driver's 20-step burst (rocprofv3 --kernel-trace), last region:  tools/trace_burst_widths.sh [bench args]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_b
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-events --steps 20 --warmup 5 "$@" > /tmp/tr_b.log 2>&1
tail -1 /tmp/tr_b.log | cut -c1-120
python $GRAFT_REPO_ROOT/tools/trace_burst.py /tmp/tr_b ${NCHAINS:-2} detail

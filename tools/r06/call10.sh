#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call10
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 4 8 6 7 15; do
rm -rf /tmp/pf10
rocprofv3 --kernel-trace -d /tmp/pf10 -o t --output-format csv -- python $REPO/bench.py --cfg5-only 1 --opt tail_dbg=$dbg > /tmp/pf10.log 2>&1
python - <<PY >> $OUT/tail_phases.txt
import csv, glob
f = glob.glob("/tmp/pf10/**/*kernel_trace.csv", recursive=True)[0]
v = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_msm_tail_fast"))
print("tail_dbg=$dbg  k_msm_tail_fast n %d median %.1f us min %.1f" % (len(v), v[len(v)//2] if v else 0, v[0] if v else 0))
PY
done
cat $OUT/tail_phases.txt

#!/bin/bash
# kernel durations of the combining queue's narrow chains under load (rocprofv3 kernel stats of the C++ client)
cd /root/repo
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=16
python tools/combine_rate.py --seconds 0.2 --window 16 "threads 1" > /dev/null 2>&1   # builds the client
cd /tmp && export TMPDIR=/tmp
for mode in "threads 1" "tickets 16 128"; do
  tag=$(echo $mode | tr ' ' '_')
  rm -rf /tmp/ct_$tag
  BP_LANES=8 BP_W=16 rocprofv3 --kernel-trace --stats -d /tmp/ct_$tag -o t --output-format csv -- /root/repo/gpurun_out/combine_rate /root/repo/bench_data/combine_rate_inputs.bin 2 $mode > /tmp/ct_$tag.log 2>&1
  echo "== $mode"; grep -E '^\{' /tmp/ct_$tag.log | cut -c1-300
  python - <<PY
import csv,glob
f=glob.glob("/tmp/ct_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0]
    if float(r['Percentage'])>0.5 and 'fb_' not in n: print("  %-36s calls %6s avg %8.1f us  min %8.1f  max %8.1f"%(n[:36],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3))
PY
done

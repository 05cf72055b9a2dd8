#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=16
R=gpurun_out/r04/combine_lanes_sweep.txt
: > $R
for L in 6 8 12 16; do for I in 4 6 8; do
  echo "# BPGPU_COMBINE_LANES=$L combine_inflight=$I" >> $R
  BPGPU_COMBINE_LANES=$L timeout 200 python tools/combine_rate.py --seconds 2 --window 16 --opts combine_inflight=$I "threads 256" "tickets 16 128" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print(d['mode'], d['threads'], d['arg'], round(d['rate_per_s']), 'p50', d['lat_ms']['p50'], 'p99', d['lat_ms']['p99'], 'ppc', d['proofs_per_chain'], 'mism', d['mismatches'])" >> $R
done; done
cat $R

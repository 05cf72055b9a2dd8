#!/bin/bash
# Round 5: kernel times of the narrow chains UNDER LOAD (16 x 128 tickets; 64 threads of single 6179-term MSM calls) under rocprofv3 --kernel-trace --stats.
set -u
cd "$(dirname "$0")/../.."
REPO=$PWD
OUT=$REPO/gpurun_out/r05i
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16 BP_LANES=8
g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,$PWD/bulletproofs_amd/csrc -o /tmp/combine_rate || exit 1
python3 tools/make_msm_inputs.py /tmp/msm_inputs.bin > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_tk; BP_W=16 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pf_tk -o t --output-format csv -- /tmp/combine_rate $REPO/bench_data/combine_rate_inputs.bin 0.8 tickets 16 128 > $OUT/tickets_under_rocprof.json 2> /dev/null
cp $(find /tmp/pf_tk -name "*kernel_stats.csv" | head -1) $OUT/tickets_16x128_kernel_stats.csv
rm -rf /tmp/pf_ms; BP_W=12 BP_MSM_INPUTS=/tmp/msm_inputs.bin timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pf_ms -o t --output-format csv -- /tmp/combine_rate $REPO/bench_data/combine_rate_inputs.bin 0.8 msm 64 1 > $OUT/msm64_under_rocprof.json 2> /dev/null
cp $(find /tmp/pf_ms -name "*kernel_stats.csv" | head -1) $OUT/msm_64_kernel_stats.csv
python3 - <<PY
import csv
for f in ("tickets_16x128", "msm_64"):
    print("==", f)
    for r in csv.DictReader(open("$OUT/%s_kernel_stats.csv" % f)):
        n = r["Name"].split("(")[0].replace("void ", "")
        if n.startswith(("k_fb_fill", "k_fb_norm", "k_fb_base", "k_from_uniform", "at::", "k_pool_spin")): continue
        print("%-26s calls %6s  avg %8.1f us  min %8.1f  max %9.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
cut -c1-200 $OUT/tickets_under_rocprof.json; cut -c1-200 $OUT/msm64_under_rocprof.json

#!/bin/bash
# GPU box: launch 1 (k_rp_stage1) before / after -- rates (interleaved A/B), the launch alone and under contention (rocprofv3 kernel
# stats), and its counters (instruction cache, scratch / VMEM instructions, HBM bytes), one --pmc pass per counter set.
#   tools/stage1_counters.sh name1 name2 ...      (ab/<name>.so)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04/stage1_ab
mkdir -p $OUT
cd $REPO
export GPU_MAX_HW_QUEUES=16
cp bulletproofs_amd/csrc/libbpgpu.so /tmp/keep.so
B="python $REPO/bench.py --no-cpu-baseline --no-extra"
ALONE="$B --direct --streams 1 --batch 5120 --steps 48 --warmup 8 --opt horner_lanes=1"
for r in 1 2 3; do
  for v in "$@"; do
    cp ab/$v.so bulletproofs_amd/csrc/libbpgpu.so
    for args in "--steps 20 --warmup 5" ""; do
      $B $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', '[$args]', round(d['value']))" >> $OUT/ab_rates.txt
    done
  done
done
cat $OUT/ab_rates.txt
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $REPO/ab/$v.so $REPO/bulletproofs_amd/csrc/libbpgpu.so
  rm -rf /tmp/s1_$v; rocprofv3 --kernel-trace --stats -d /tmp/s1_$v -o t --output-format csv -- $ALONE > /tmp/s1_$v.log 2>&1
  cp $(find /tmp/s1_$v -name "*kernel_stats.csv" | head -1) $OUT/${v}_alone_kernel_stats.csv
  rm -rf /tmp/s2_$v; rocprofv3 --kernel-trace --stats -d /tmp/s2_$v -o t --output-format csv -- $B > /tmp/s2_$v.log 2>&1
  cp $(find /tmp/s2_$v -name "*kernel_stats.csv" | head -1) $OUT/${v}_default_kernel_stats.csv
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
    i=$((i+1)); rm -rf /tmp/pm_${v}_$i
    rocprofv3 --kernel-trace --pmc $set -d /tmp/pm_${v}_$i -o t --output-format csv -- $B --direct --streams 1 --batch 5120 --steps 6 --warmup 2 --opt horner_lanes=1 > /tmp/pm_${v}_$i.log 2>&1 || echo "pmc set '$set' failed for $v" >> $OUT/pmc_errors.txt
  done
  python - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pm_${v}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("$OUT/${v}_counters_per_launch.json", "w"), indent=1)
for k, cs in sorted(out.items()):
    if "stage1" in k or "rp_points" in k:
        print("$v", k, {c: round(x) for c, x in cs.items()})
PY
done
cp /tmp/keep.so $REPO/bulletproofs_amd/csrc/libbpgpu.so
for v in "$@"; do echo "== $v alone"; grep -E "stage1|stage4|horner|stage3|finish" $OUT/${v}_alone_kernel_stats.csv | cut -d, -f1-4,8 | head -8; echo "== $v default (contended)"; grep -E "stage1" $OUT/${v}_default_kernel_stats.csv | cut -d, -f1-4,8; done
cat $OUT/pmc_errors.txt 2>/dev/null

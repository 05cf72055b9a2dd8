#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the default bench command, plus separate
# --pmc passes for HBM traffic (FETCH_SIZE / WRITE_SIZE), written under gpurun_out/prof_$1/.
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf /tmp/pm1 /tmp/pm2
rocprofv3 --kernel-trace --stats -d /tmp/pf -o t --output-format csv -- python $REPO/bench.py --no-cpu-baseline > /tmp/pf.log 2>&1
cp $(find /tmp/pf -name "*kernel_stats.csv" | head -1) $OUT/bench_default_kernel_stats.csv
grep '"metric"' /tmp/pf.log > $OUT/bench_default_under_rocprof.json
rocprofv3 --kernel-trace --stats -d /tmp/pf1 -o t --output-format csv -- python $REPO/bench.py --no-cpu-baseline --streams 1 > /tmp/pf1.log 2>&1
cp $(find /tmp/pf1 -name "*kernel_stats.csv" | head -1) $OUT/bench_streams1_kernel_stats.csv
grep '"metric"' /tmp/pf1.log > $OUT/bench_streams1_under_rocprof.json
# counters in their own passes (kernel-trace only, no other trace domains)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pm1 -o t --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 8 --warmup 2 --streams 1 > /tmp/pm1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pm2 -o t --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 8 --warmup 2 --streams 1 > /tmp/pm2.log 2>&1
python - <<PY
import csv, collections, json, glob
out = {}
for d, name in (("/tmp/pm1", "FETCH_SIZE"), ("/tmp/pm2", "WRITE_SIZE")):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0]
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    out[name] = {k: {"dispatches": v[0], "avg_per_dispatch": v[1] / v[0]} for k, v in acc.items()}
json.dump(out, open("$OUT/pmc_fetch_write_raw.json", "w"), indent=1)
for name in out:
    for k, v in sorted(out[name].items(), key=lambda kv: -kv[1]["avg_per_dispatch"])[:12]:
        print(name, k, v)
PY

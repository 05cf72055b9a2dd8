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
# VALU instruction mix / work per launch (tools/pmc_insts.sh writes gpurun_out/pmc_insts/)
$REPO/tools/pmc_insts.sh > $OUT/pmc_insts.log 2>&1; cp $REPO/gpurun_out/pmc_insts/insts.json $OUT/pmc_instruction_mix_streams1.json; cp $REPO/gpurun_out/pmc_insts/valu_work_cfg2.json $OUT/valu_work_cfg2.json
python - <<PY
import csv, collections, json, glob, re
def short(name):
    n = name.split("(")[0].strip()
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"<.*$", "", n)
    return n[2:] if n.startswith("k_") else n
out = {}
for d, name in (("/tmp/pm1", "FETCH_SIZE"), ("/tmp/pm2", "WRITE_SIZE")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = short(r["Kernel_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    out[name] = {k: {"dispatches": v[0], "avg_per_dispatch": v[1] / v[0]} for k, v in acc.items()}
json.dump(out, open("$OUT/pmc_fetch_write_raw.json", "w"), indent=1)
# HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE is in KB and counts a
# 128-byte request as 64 bytes (x2); WRITE_SIZE is in KB
traffic = {"_note": "HBM bytes per launch = 2*FETCH_SIZE[KB]*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B, "
           "MI355X_MICROARCH.md section HBM) + WRITE_SIZE[KB]*1024; separate --pmc passes of "
           "bench.py --steps 8 --warmup 2 --streams 1 (cfg2, batch 1024), tools/collect_profiles.sh; raw counters in "
           "profiles/$TAG/pmc_fetch_write_raw.json"}
for k in out["FETCH_SIZE"]:
    rd = out["FETCH_SIZE"][k]["avg_per_dispatch"]; wr = out["WRITE_SIZE"].get(k, {"avg_per_dispatch": 0})["avg_per_dispatch"]
    traffic[k] = int(2 * rd * 1024 + wr * 1024)
json.dump(traffic, open("$OUT/pmc_traffic_cfg2.json", "w"), indent=1)
for k, v in sorted(((k, v) for k, v in traffic.items() if k != "_note"), key=lambda kv: -kv[1])[:14]:
    print("HBM bytes/launch %-28s %14d" % (k, v))
PY

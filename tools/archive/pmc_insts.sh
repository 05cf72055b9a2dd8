#!/bin/bash
# GPU box: per-kernel instruction mix and wave cycles (rocprofv3 --pmc, kernel-trace only) for the single-stream bench.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_insts
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pi$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pi$i -o t --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 6 --warmup 2 --streams 1 $EXTRA > /tmp/pi$i.log 2>&1
done
python - <<PY
import csv, collections, json, glob
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in glob.glob("/tmp/pi*/"):
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            a = acc[k][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("$OUT/insts.json", "w"), indent=1)
# VALU work per launch, by short kernel name (what bench.py reads as profiles/valu_work_cfg2.json)
import re
def short(name):
    n = re.sub(r"^void\s+", "", name.strip()); n = re.sub(r"<.*$", "", n)
    return n[2:] if n.startswith("k_") else n
work = {"_note": "SQ_INSTS_VALU per launch (wavefront-instructions) of bench.py --streams 1 at cfg2, batch 1024 "
        "(rocprofv3 --pmc, tools/pmc_insts.sh); one batch = one launch of each rp_*/fb_reduce/finish8 kernel "
        "(fb_reduce: its per-launch average x 1 launch)"}
for k, cs in out.items():
    if "SQ_INSTS_VALU" in cs: work[short(k)] = int(cs["SQ_INSTS_VALU"])
json.dump(work, open("$OUT/valu_work_cfg2.json", "w"), indent=1)
for k, cs in out.items():
    w = cs.get("SQ_WAVES", 0) or 1
    print("%-16s waves %6d  valu/wave %8.0f salu/wave %7.0f lds/wave %6.0f vmem/wave %5.0f  wave_cycles/wave %9.0f  gui_active %8.0f" % (
        k[:16], w, cs.get("SQ_INSTS_VALU", 0) / w, cs.get("SQ_INSTS_SALU", 0) / w, cs.get("SQ_INSTS_LDS", 0) / w,
        (cs.get("SQ_INSTS_VMEM_RD", 0) + cs.get("SQ_INSTS_VMEM_WR", 0)) / w, cs.get("SQ_WAVE_CYCLES", 0) / w, cs.get("GRBM_GUI_ACTIVE", 0)))
PY

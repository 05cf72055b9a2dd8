#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the bench commands, plus separate --pmc passes
# (kernel-trace only, no other trace domains) for HBM traffic (FETCH_SIZE / WRITE_SIZE) and VALU work
# (SQ_INSTS_VALU) per kernel, for cfg2 (the headline), cfg3 and the cfg5 MSM shape.  Output: gpurun_out/prof_$1/.
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-extra"

stats() {  # name, args...: kernel stats csv + the bench line measured under the profiler
  local name=$1; shift
  rm -rf /tmp/pf_$name
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -o t --output-format csv -- "$@" > /tmp/pf_$name.log 2>&1
  cp $(find /tmp/pf_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
  grep -E '^\{' /tmp/pf_$name.log > $OUT/${name}_under_rocprof.json
}
pmc() {  # name, counter list (quoted), args...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pm_$name
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm_$name -o t --output-format csv -- "$@" > /tmp/pm_$name.log 2>&1
}

if [ -z "$ONLY_PMC" ]; then
stats bench_default $B
stats bench_steps20 $B --steps 20 --warmup 5
stats bench_streams1 $B --direct --streams 1 --steps 256 --warmup 16
stats bench_wide_chain_alone $B --direct --streams 1 --batch 5120 --steps 64 --warmup 8 --opt horner_lanes=1
stats bench_cfg3 $B --config cfg3 --steps 640 --warmup 64
stats bench_cfg5 python $REPO/bench.py --cfg5-only 8

# the library's scheduler: burst / steady / one-call rates; the constant-time prover's counters
python $REPO/tools/pool_rate.py burst steady host > $OUT/pool_rate.txt 2>&1
bash $REPO/tools/ct_counters.sh > $OUT/prover_constant_time_counters.txt 2>&1
$REPO/tools/microbench_gather > $OUT/microbench_gather.txt 2>&1
# the two bench lines as the driver runs them (not under the profiler)
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
fi

# counters of the WIDE chain form (what the pool issues: >= 2048 proofs per launch chain -- one-lane Horner chain aside, window sums as
# their own launch, A outside the window sums), one stream, no pool
# counters of the WIDE chain form, issued the way the pool issues it (>= 2048 proofs per launch chain, other chains assumed beside it: one-lane
# Horner chain aside on the second stream, window sums as their own launch, A outside the window sums): K steps through a one-lane pool =
# ONE coalesced chain per region; latency_proofs=0 tells the pool not to treat the lone chain as alone (it would take the latency forms)
declare -A WSTEPS=( [cfg2]=5 [cfg3]=8 [cfg4]=4 )    # x batch 1024 / 256 / 512 = 5120 / 2048 / 2048 proofs per chain
PMC_CFGS=${PMC_CFGS:-"cfg2 cfg3 cfg4 cfg5"}   # (e.g. PMC_CFGS=cfg5 ONLY_PMC=1: re-collect one configuration's counters)
for cfg in cfg2 cfg3 cfg4; do
  case " $PMC_CFGS " in *" $cfg "*) ;; *) continue;; esac
  A="$B --config $cfg --steps ${WSTEPS[$cfg]} --warmup 0 --streams 1 --repeat 3 --opt latency_proofs=0,auto_flush_items=64"
  pmc ${cfg}_fetch FETCH_SIZE $A
  pmc ${cfg}_write WRITE_SIZE $A
  pmc ${cfg}_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" $A
done
case " $PMC_CFGS " in *" cfg5 "*)
pmc cfg5_fetch FETCH_SIZE python $REPO/bench.py --cfg5-only 1
pmc cfg5_write WRITE_SIZE python $REPO/bench.py --cfg5-only 1
pmc cfg5_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" python $REPO/bench.py --cfg5-only 1
;; esac

python $REPO/tools/isa_mix.py > /tmp/isa_mix.json
python - <<PY
import csv, collections, json, glob, re
LABEL = {"vb_window_wide": "rp_stage3w", "vb_window_colc": "rp_stage3w", "rp_horner_wide": "rp_horner1", "rp_exponents": "rp_stage3", "rp_exponents_w3": "rp_stage3",
         "bk2_prepare": "bk_prepare", "bk2_window": "bk_window", "bk2_leafv": "bk_leaf_narrow", "msm_tail_fast": "msm_tail_narrow", "fb_walk1": "fb_walk_narrow"}   # kernel -> the library's launch label (bench.py's names)
def short(name):
    n = name.split("(")[0].strip()
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"<.*$", "", n)
    n = n[2:] if n.startswith("k_") else n
    return LABEL.get(n, n)
def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    # per kernel only the dispatches of its LARGEST grid: the bench commands also issue narrow launches of the same kernels (a lone MSM,
    # a latency probe), and an average over both is the figure of neither (rounds 2-5 did that for cfg5: 3.58 M wavefront-instructions
    # per MSM reported where the batch of 64 executed 5.0 M)
    rows = collections.defaultdict(lambda: collections.defaultdict(list))   # label -> (kernel variant, grid) -> values
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        full = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
        rows[short(r["Kernel_Name"])][(full, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    out = {}
    for k, groups in rows.items():
        # the chain's own launches of this label: the (variant, grid) group with the largest value per dispatch (a warm-up chain of one
        # proof, a latency probe and a lone MSM launch the same kernels -- or their narrow-chain variants -- at other grids)
        (full, g), w = max(groups.items(), key=lambda kv: sum(kv[1]) / len(kv[1]))
        out[k] = {"dispatches": len(w), "avg_per_dispatch": sum(w) / len(w), "grid": g, "variant": full}
    return out
for cfg, what, ppl in (("cfg2", "bench.py --config cfg2 --steps 5 --warmup 0 --streams 1 --opt latency_proofs=0,auto_flush_items=64 (the pool's wide chain form: one coalesced chain of 5120 per region)", 5120),
                  ("cfg3", "bench.py --config cfg3 --steps 8 --warmup 0 --streams 1 --opt latency_proofs=0,auto_flush_items=64 (one coalesced chain of 2048 per region)", 2048),
                  ("cfg4", "bench.py --config cfg4 --steps 4 --warmup 0 --streams 1 --opt latency_proofs=0,auto_flush_items=64 (one coalesced chain of 2048 per region)", 2048),
                  ("cfg5", "bench.py --cfg5-only 1 (batches of 64 MSMs of 6179 terms)", 64)):
    if cfg not in "$PMC_CFGS".split(): continue
    rd, wr = per_kernel("/tmp/pm_%s_fetch" % cfg, "FETCH_SIZE"), per_kernel("/tmp/pm_%s_write" % cfg, "WRITE_SIZE")
    json.dump({"FETCH_SIZE": rd, "WRITE_SIZE": wr}, open("$OUT/pmc_fetch_write_raw_%s.json" % cfg, "w"), indent=1)
    # HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE is in KB and counts a
    # 128-byte request as 64 bytes (x2); WRITE_SIZE is in KB
    traffic = {"_note": "HBM bytes per launch = 2*FETCH_SIZE[KB]*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md "
               "section HBM) + WRITE_SIZE[KB]*1024; separate rocprofv3 --pmc passes of %s, tools/collect_profiles.sh; raw counters in "
               "profiles/$TAG/pmc_fetch_write_raw_%s.json" % (what, cfg), "_proofs_per_launch": ppl}
    for k in rd:
        traffic[k] = int(2 * rd[k]["avg_per_dispatch"] * 1024 + wr.get(k, {"avg_per_dispatch": 0})["avg_per_dispatch"] * 1024)
    json.dump(traffic, open("$OUT/pmc_traffic_%s.json" % cfg, "w"), indent=1)
    va = per_kernel("/tmp/pm_%s_valu" % cfg, "SQ_INSTS_VALU")
    work = {"_note": "SQ_INSTS_VALU per launch (wavefront-instructions), rocprofv3 --pmc pass of %s, tools/collect_profiles.sh; one chain = one launch of "
            "each rp_* / finish8 kernel.  _mad_u64_fraction: share of v_mad_u64_u32 among the VALU instructions of each kernel's SHIPPED machine code "
            "(static count over the disassembly, tools/isa_mix.py), weighted by the kernels' SQ_INSTS_VALU" % what, "_proofs_per_launch": ppl}
    for k in va: work[k] = int(va[k]["avg_per_dispatch"])
    mix = json.load(open("/tmp/isa_mix.json"))
    per_k, num, den = {}, 0.0, 0.0
    PREF = {"rp_stage1": "k_rp_stage1<true>", "rp_stage4": "k_rp_stage4<4>", "rp_stage3w": "k_vb_window_wide<false>", "rp_horner1": "k_rp_horner_wide<false>",
            "finish8": "k_finish8<false>", "rp_stage3": "k_rp_exponents"}   # the variants a wide chain of the pool runs
    for full, mv in mix.items():
        lab = short(full)
        if lab in work and (lab not in per_k or PREF.get(lab) == re.sub(r"^void\\s+", "", full)):
            per_k[lab] = mv["fraction"]
    per_k = {k: fr for k, fr in per_k.items() if k.startswith(("rp_", "finish", "fb_reduce", "fb_walk", "msm_tail", "vb_", "bk_", "rlc_"))}   # the chain's kernels, not the table construction
    for k, fr in per_k.items():
        num += fr * work[k]; den += work[k]
    work["_mad_u64_fraction"] = round(num / den, 4) if den else 0.58
    work["_mad_u64_fraction_per_kernel"] = per_k
    json.dump(work, open("$OUT/valu_work_%s.json" % cfg, "w"), indent=1)
    print(cfg, "HBM bytes/launch:", {k: v for k, v in sorted(traffic.items(), key=lambda kv: -kv[1] if isinstance(kv[1], int) else 0)[:8] if not k.startswith("_")})
PY

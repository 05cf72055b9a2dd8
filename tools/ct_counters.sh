#!/bin/bash
# Evidence for the prover's constant-time option: SQ instruction / memory-request counters per kernel of one
# bpgpu_rangeproof_prove_batch call (64 proofs of (64, 1)) for three different secret sets, option on and off
# (separate rocprofv3 --pmc passes, kernel trace only).  Output: a table on stdout.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ct_probe.py <<PY
import hashlib, sys
sys.path.insert(0, "$REPO")
import bulletproofs_amd as bp
tag = sys.argv[1].encode()
n, m, nb = 64, 1, 64
vals = [int.from_bytes(hashlib.shake_256(b"%s-v%d" % (tag, i)).digest(8), "little") for i in range(nb * m)]
if tag == b"zeros":
    vals = [0] * (nb * m)
bl = b"".join(hashlib.shake_256(b"%s-b%d" % (tag, i)).digest(31) + b"\x00" for i in range(nb * m))
rng = hashlib.shake_256(tag + b"-rng").digest(64 * (m * (2 * n + 2) + 2 * m) * nb)
ctx = bp.Context(0, fixed_window_bits=12)
ctx.set_option("prover_constant_time", int(sys.argv[2]))
ctx.gens_create(n, m)
ctx.rangeproof_prove_batch(n, m, vals, bl, label=b"ct-probe", rng=rng)
ctx.close()
PY
for ct in 1 0; do
  for tag in alpha beta zeros; do
    for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
      d=/tmp/ctc_${ct}_${tag}_$(echo $ctr | cut -d' ' -f1); rm -rf $d
      rocprofv3 --kernel-trace --pmc $ctr -d $d -o t --output-format csv -- python /tmp/ct_probe.py $tag $ct > /dev/null 2>&1
    done
  done
done
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for ct in (1, 0):
    for tag in ("alpha", "beta", "zeros"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for d in glob.glob("/tmp/ctc_%d_%s_*" % (ct, tag)):
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, cs in acc.items():
            rows[(ct, k)][tag] = cs
print("prover_constant_time | kernel | counter | secrets alpha | beta | all-zero values | identical")
for (ct, k) in sorted(rows, key=lambda x: (-x[0], x[1])):
    if not k.startswith(("k_rpp", "k_fb", "k_shared", "k_ippc", "k_vb", "k_bk", "k_horner")):
        continue
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"):
        v = [rows[(ct, k)].get(t, {}).get(c, float("nan")) for t in ("alpha", "beta", "zeros")]
        print("%d | %s | %s | %d | %d | %d | %s" % (ct, k, c, v[0], v[1], v[2], "yes" if v[0] == v[1] == v[2] else "NO"))
PY

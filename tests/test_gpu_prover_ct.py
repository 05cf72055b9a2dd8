"""The constant-time option of the batched prover (context option "prover_constant_time"): the secret-dependent commitments
V_j, A, S, T_1, T_2 -- which the reference computes with its constant-time multiscalar_mul (src/range_proof/party.rs:99-124,
179-187; src/generators.rs:39-41) -- go through a small-window table walk whose addresses and instruction stream do not depend
on the scalars (csrc/msm_fixed.h fb_accum_ct_thread).  Proofs must be byte-identical to the variable-time path and to the
oracle's prover; the executed-instruction and memory-request counts must not depend on the secrets."""
import csv
import glob
import hashlib
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(nb, m, tag):
    vals = [int.from_bytes(hashlib.shake_256(b"%s-v%d" % (tag, i)).digest(8), "little") for i in range(nb * m)]
    bl = b"".join(hashlib.shake_256(b"%s-b%d" % (tag, i)).digest(31) + b"\x00" for i in range(nb * m))
    return vals, bl


@pytest.mark.parametrize("n,m,nb", [(64, 1, 70), (32, 4, 9), (8, 2, 130)])
def test_constant_time_proofs_are_byte_identical(oracle, n, m, nb):
    import bulletproofs_amd as bp
    vals, bl = _inputs(nb, m, b"ct")
    vals = [v % (1 << n) for v in vals]
    per = 64 * (m * (2 * n + 2) + 2 * m)
    seeds = [b"ct-%d-%d-%d" % (n, m, p) for p in range(nb)]
    rng = b"".join(hashlib.shake_256(sd).digest(per) for sd in seeds)       # the draws the oracle's prover derives from these seeds
    out = {}
    for ct in (0, 1):
        ctx = bp.Context(0, fixed_window_bits=12)
        ctx.set_option("prover_constant_time", ct)
        ctx.gens_create(n, m)
        out[ct] = ctx.rangeproof_prove_batch(n, m, vals, bl, label=b"ct-test", rng=rng)
        if ct:
            assert ctx.get_option("prover_constant_time") == 1
            assert ctx.get_option("staging_residue") == 0          # secrets cleared on the way out (prover_exit)
            pl = len(out[ct][0]) // nb
            assert ctx.rangeproof_verify_batch(n, m, out[ct][0], pl, out[ct][1], b"ct-test") == bytes(nb)
        ctx.close()
    assert out[0] == out[1]
    # and the oracle's prover makes the same bytes from the same draws (first proofs of the batch)
    g = oracle.Gens(n, m)
    pl = len(out[1][0]) // nb
    st0 = oracle.transcript_new(b"ct-test")
    for i in range(min(nb, 3)):
        pr, cm, _ = oracle.prove_ts(g, vals[i * m:(i + 1) * m], bl[32 * m * i:32 * m * (i + 1)], n, st0, seeds[i])
        assert pr == out[1][0][pl * i:pl * (i + 1)] and cm == out[1][1][32 * m * i:32 * m * (i + 1)]


_PROBE = r'''
import hashlib, sys
sys.path.insert(0, %r)
import bulletproofs_amd as bp
tag = sys.argv[1].encode()
n, m, nb = 64, 1, 64
vals = [int.from_bytes(hashlib.shake_256(b"%%s-v%%d" %% (tag, i)).digest(8), "little") for i in range(nb * m)]
if tag == b"zeros":
    vals = [0] * (nb * m)                        # every a_L bit 0: the extreme case for a digit-dependent walk
bl = b"".join(hashlib.shake_256(b"%%s-b%%d" %% (tag, i)).digest(31) + b"\x00" for i in range(nb * m))
rng = hashlib.shake_256(tag + b"-rng").digest(64 * (m * (2 * n + 2) + 2 * m) * nb)
ctx = bp.Context(0, fixed_window_bits=12)
ctx.set_option("prover_constant_time", int(sys.argv[2]))
ctx.gens_create(n, m)
ctx.rangeproof_prove_batch(n, m, vals, bl, label=b"ct-probe", rng=rng)
ctx.close()
'''


def _counters(tmp, tag, ct, counters):
    """per-kernel sums of the given SQ counters of one prove call (rocprofv3 --pmc, kernel trace only)"""
    d = os.path.join(tmp, "pmc_%s_%d_%s" % (tag, ct, counters[0]))
    script = os.path.join(tmp, "probe.py")
    with open(script, "w") as f:
        f.write(_PROBE % ROOT)
    env = dict(os.environ, TMPDIR=tmp)
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", d, "-o", "t", "--output-format", "csv", "--", sys.executable, script, tag, str(ct)],
                          cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


@pytest.mark.skipif(shutil.which("rocprofv3") is None, reason="rocprofv3 not on PATH")
def test_constant_time_counters_do_not_depend_on_the_secrets(tmp_path):
    """SQ_INSTS_VALU / SQ_INSTS_SALU and SQ_INSTS_VMEM_RD / SQ_INSTS_SMEM of every kernel up to the inner-product argument are
    the same for three different secret sets (random, other random, all-zero values) with the option on; with the option off the
    table walk's counts differ (its zero digits skip additions) -- the control that the probe can see a leak."""
    tmp = str(tmp_path)
    # the kernels that touch the secrets before the (deliberately variable-time, as in the reference: ipp.rs:87-178) inner-product rounds
    secret_kernels = ("k_rpp_commit1", "k_fb_recode_ct", "k_fb_accum_ct", "k_fb_reduce", "k_shared_finish", "k_rpp_chal1", "k_rpp_poly", "k_rpp_tcommit", "k_rpp_chal2")
    for counters in (["SQ_INSTS_VALU", "SQ_INSTS_SALU"], ["SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"]):
        runs = [_counters(tmp, tag, 1, counters) for tag in ("alpha", "beta", "zeros")]
        seen = 0
        for k in runs[0]:
            if not any(k.startswith(s) for s in secret_kernels):
                continue
            seen += 1
            assert runs[0][k] == runs[1][k] == runs[2][k], (k, runs[0][k], runs[1][k], runs[2][k])
        assert seen >= 6, sorted(runs[0])
    var = [_counters(tmp, tag, 0, ["SQ_INSTS_VALU", "SQ_INSTS_SALU"]) for tag in ("alpha", "zeros")]
    walk = [k for k in var[0] if k.startswith("k_fb_accum")]
    assert walk and any(var[0][k] != var[1][k] for k in walk)

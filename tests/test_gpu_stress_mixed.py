"""Every kind of entry point mixed on the same contexts from several host threads: range-proof verification (host pointers and
submit / collect), batch-combined verification, linear-proof verification (explicit bases and generator tables), linear-proof
and range-proof creation, stand-alone inner-product verification, share audits, plain MSMs.  All of them share a context's
arena, its working-set block, its device IO block and its pinned staging buffer, at different sizes from call to call -- the
test that would catch a stale pointer after one of those blocks grows, or a call reading another call's scratch.  Expected
results are computed once, up front, by the oracle."""
import hashlib
import os
import random
import threading

import pytest

pytestmark = pytest.mark.gpu


def test_mixed_entry_points_from_threads(oracle):
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    g = oracle.Gens(64, 2)
    # ---- inputs and expected outputs
    fx = wl.load_fixture("cfg2_n64_m1")
    nb = 40
    rp_proofs = bytearray(fx.proofs[:nb * fx.proof_len])
    for i in (3, 17, 29):
        rp_proofs[i * fx.proof_len + 130 + i] ^= 2
    rp_proofs, rp_coms = bytes(rp_proofs), fx.commitments[:nb * 32]
    rp_rng = hashlib.shake_256(b"mix-rng").digest(64 * nb)
    _, rp_expect, rp_msm = oracle.verify_batch(g, rp_proofs, rp_coms, 1, 64, fx.label, rp_rng, threads=os.cpu_count() or 1)
    lin = [oracle.linear_test_instance(16, b"mix-lin-%d" % j) for j in range(6)]
    lt = bytearray(lin[2]["proof"]); lt[-40] ^= 1
    lin[2] = dict(lin[2], proof=bytes(lt))
    lcat = lambda key: b"".join(i[key] for i in lin)
    st_lin = oracle.transcript_new(lin[0]["label"])
    lin_expect = [oracle.linear_verify(16, i["proof"], st_lin, i["C"], i["G"], i["F"], i["B"], i["b"]) for i in lin]
    lin_made = [oracle.linear_create(16, st_lin, i["rng"], i["C"], i["r"], i["a"], i["b"], i["G"], i["F"], i["B"])[1] for i in lin]
    ipp = [oracle.ipp_test_instance(32, b"innerproducttest", b"mix-ipp-%d" % j) for j in range(5)]
    ipp[1] = dict(ipp[1], P=ipp[1]["Q"])
    icat = lambda key: b"".join(i[key] for i in ipp)
    ipp_expect = [oracle.ipp_verify(32, i["proof"], b"innerproducttest", i["Gf"], i["Hf"], i["P"], i["Q"], i["G"], i["H"]) for i in ipp]
    sh = oracle.prove_shares(g, [77, (1 << 40) + 5], bytes(64), 32, b"mix-mpc", b"s")          # party 1 commits to a 41-bit value at n = 32
    aud_expect = [oracle.audit_share(g, 32, j, sh["shares"][32 * 67 * j:32 * 67 * (j + 1)], sh["bit_commitments"][96 * j:96 * j + 96],
                                     sh["poly_commitments"][64 * j:64 * j + 64], sh["challenges"])[0] for j in range(2)]
    assert aud_expect == [0, 1]
    msm_sc = hashlib.shake_256(b"mix-msm").digest(32 * 300)
    msm_sc = b"".join((int.from_bytes(msm_sc[32 * i:32 * i + 32], "little") >> 4).to_bytes(32, "little") for i in range(300))
    Gc, Hc, _, _ = g.export()
    msm_pts = (Gc + Hc)[:32 * 128] * 3
    msm_pts = msm_pts[:32 * 300]
    msm_expect = [oracle.msm(msm_sc[:32 * 7], msm_pts[:32 * 7])[1], oracle.msm(msm_sc[32 * 7:], msm_pts[32 * 7:])[1]]
    pv = [int.from_bytes(hashlib.shake_256(b"mix-v%d" % i).digest(4), "little") for i in range(6)]
    pbl = b"".join(hashlib.shake_256(b"mix-b%d" % i).digest(31) + b"\x00" for i in range(6))
    # ---- contexts: two private ones and one shared by two threads
    ctxs = [bp.Context(0) for _ in range(3)]
    for c in ctxs:
        c.gens_create(64, 2)
    errors, done = [], []

    def check(cond, what):
        done.append(what)
        if not cond:
            errors.append(what)

    def worker(tid, c, iters, shared=False):
        rnd = random.Random(1000 + tid)
        try:
            for it in range(iters):
                op = rnd.randrange(9)
                if shared and op == 1:
                    op = 0            # a submit / collect pair is one caller's business: not interleaved by two threads on one context
                if op == 0:
                    v, mo = c.rangeproof_verify_batch(64, 1, rp_proofs, fx.proof_len, rp_coms, fx.label, rp_rng, want_msm=True)
                    check(v == rp_expect and mo == rp_msm, "rp verify")
                elif op == 1:
                    k = rnd.randrange(1, nb)
                    c.rangeproof_verify_batch_submit(64, 1, rp_proofs[:k * fx.proof_len], fx.proof_len, rp_coms[:32 * k], fx.label, rp_rng[:64 * k])
                    v = c.collect()
                    check(bytes(v[0] if isinstance(v, tuple) else v)[:k] == rp_expect[:k], "rp submit/collect")
                elif op == 2:
                    v, _, _ = c.rangeproof_verify_rlc(64, 1, rp_proofs, fx.proof_len, rp_coms, fx.label, rp_rng, None)
                    check(bytes(v) == rp_expect, "rp rlc")
                elif op == 3:
                    fixed = rnd.random() < 0.5
                    a = (None, None, None) if fixed else (lin[0]["G"], lin[0]["F"], lin[0]["B"])
                    v, mo = c.linear_verify_batch(16, lcat("proof"), len(lin[0]["proof"]), lcat("C"), *a, lcat("b"), label=lin[0]["label"], want_msm=True)
                    check(list(v) == [e[0] for e in lin_expect] and mo[:32] == lin_expect[0][1] and mo[64:96] == lin_expect[2][1], "linear verify")
                elif op == 4:
                    fixed = rnd.random() < 0.5
                    a = (None, None, None) if fixed else (lin[0]["G"], lin[0]["F"], lin[0]["B"])
                    pr, stt = c.linear_create_batch(16, lcat("C"), lcat("r"), lcat("a"), lcat("b"), *a, label=lin[0]["label"], rng=lcat("rng"))
                    check(stt == bytes(6) and pr == b"".join(lin_made), "linear create")
                elif op == 5:
                    v, mo = c.ipp_verify_batch(32, icat("proof"), len(ipp[0]["proof"]), b"innerproducttest", icat("Gf"), icat("Hf"), icat("P"), icat("Q"),
                                               icat("G"), icat("H"), want_msm=True)
                    check(list(v) == [e[0] for e in ipp_expect] and mo[32:64] == ipp_expect[1][1], "ipp verify")
                elif op == 6:
                    v = c.rangeproof_audit_shares(32, [0, 1], sh["shares"], sh["bit_commitments"], sh["poly_commitments"], sh["challenges"])
                    check(list(v) == aud_expect, "audit")
                elif op == 7:
                    out, stt = c.msm_batch([7, 293], msm_sc, msm_pts)
                    check(stt == bytes(2) and out == msm_expect[0] + msm_expect[1], "msm")
                else:
                    proofs, coms = c.rangeproof_prove_batch(32, 2, pv, pbl, label=b"mix-prove")
                    pl = len(proofs) // 3
                    check(c.rangeproof_verify_batch(32, 2, proofs, pl, coms, b"mix-prove") == bytes(3), "prove/verify")
        except Exception as e:   # noqa: BLE001
            errors.append("thread %d: %r" % (tid, e))

    ths = [threading.Thread(target=worker, args=(0, ctxs[0], 150)), threading.Thread(target=worker, args=(1, ctxs[1], 150)),
           threading.Thread(target=worker, args=(2, ctxs[2], 120, True)), threading.Thread(target=worker, args=(3, ctxs[2], 120, True))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for c in ctxs:
        c.close()
    assert not errors, errors[:5]
    assert len(done) == 2 * 150 + 2 * 120 and all(sum(1 for d in done if d == w) >= 20 for w in set(done)) and len(set(done)) == 9

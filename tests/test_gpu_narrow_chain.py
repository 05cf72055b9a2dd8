"""The narrow chain's round-6 forms (chains of up to 256 proofs; csrc/k_rp1.hip k_rp_stage1_coop, k_rp34.hip): every option that selects
one -- coop_split, exp_single, narrow_chunk, narrow_walk, narrow_hi_max, narrow_hi4_max, narrow_fused_finish -- flipped on its own and all together, against the default
build of the chain AND the oracle: verdicts, mega-check encodings and advanced transcripts must be bit-identical whatever the form
(src/range_proof/mod.rs:345-452 is one function; how the device cuts it into lanes is nobody's business but ours)."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu

FORMS = [
    {"coop_split": 0, "exp_single": 0, "narrow_chunk": 32, "narrow_walk": 0, "narrow_hi_max": 0},   # round 5's chain
    {"coop_split": 1, "narrow_hi_max": 0},
    {"exp_single": 0, "narrow_hi_max": 0},
    {"narrow_chunk": 3, "narrow_hi_max": 0},
    {"narrow_chunk": 32, "narrow_hi_max": 0},
    {"narrow_walk": 0, "narrow_hi_max": 0},
    {"narrow_hi_max": 0},
    {"narrow_hi_max": 4},      # (batches of 5 and more take the 64-window chain)
    {"narrow_hi_max": 256},
    {"narrow_hi_max": 256, "narrow_chunk": 5},
    {"coop_split": 0, "narrow_hi_max": 256},   # (second tables need the split launch 1: falls back to 64 windows)
    {"narrow_hi4_max": 256},                   # (chains of up to narrow_hi_max = 32 proofs then take four table levels and a 16-window chain)
    {"narrow_hi4_max": 4, "narrow_chunk": 6},
    {"narrow_hi4_max": 256, "narrow_hi_max": 256, "narrow_fused_finish": 0},
    {"narrow_fused_finish": 0},
    {"narrow_fused_finish": 0, "narrow_hi_max": 0},
]


@pytest.fixture(scope="module")
def contexts():
    import bulletproofs_amd as bp
    out = []
    for f in FORMS:
        c = bp.Context(0)
        for k, v in f.items():
            c.set_option(k, v)
        c.gens_create(64, 8)
        out.append(c)
    yield out
    for c in out:
        c.close()


def test_every_form_on_golden_shapes_and_ragged_widths(contexts, oracle, oracle_gens_64_8, golden):
    label, vc = golden["label"], golden["vc_bytes"]
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        pr = bytes.fromhex(case["proof"])
        bad = bytearray(pr)
        bad[128] ^= 1                      # wrong t_x
        fmt = bytearray(pr)
        fmt[160:192] = b"\xff" * 32        # FormatError
        ident = bytearray(pr)
        ident[224:256] = bytes(32)         # L_0 = identity encoding
        undec = bytearray(pr)
        undec[32] |= 1                     # S: a non-canonical (odd) encoding -- the decode role rejects it, the second-table role must not mind
        for nb in (1, 2, 5, 33):
            batch = [[pr, bytes(bad), bytes(fmt), bytes(ident), bytes(undec), pr][i % 6] for i in range(nb)]
            proofs, coms = b"".join(batch), vc[:32 * m] * nb
            rng = hashlib.shake_256(b"narrow-%d-%d-%d" % (n, m, nb)).digest(64 * nb)
            res = [c.rangeproof_verify_batch(n, m, proofs, len(pr), coms, label, rng, want_msm=True) for c in contexts]
            for i, r in enumerate(res[1:]):
                assert r == res[0], (n, m, nb, FORMS[i + 1])
            assert list(res[0][0]) == [[0, 1, 2, 1, 1, 0][i % 6] for i in range(nb)]
            # verdicts only (the crate's call: the last workgroup of a proof in launch 4 finishes it), twice in a row on every context
            for rep in range(2):
                for i, c in enumerate(contexts):
                    v = c.rangeproof_verify_batch(n, m, proofs, len(pr), coms, label, rng)
                    v = v[0] if isinstance(v, tuple) else v
                    assert bytes(v) == bytes(res[0][0]), (n, m, nb, rep, FORMS[i])
            if nb == 2:
                rc, enc = oracle.verify(oracle_gens_64_8, bytes(bad), vc[:32 * m], n, label, rng[64:128])[:2]
                assert rc == 1 and enc == res[-2][1][32:64]


def test_every_form_with_caller_transcripts_and_the_bench_shape(contexts, oracle, oracle_gens_64_8):
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg2_n64_m1")
    for nb in (1, 7, 32, 33, 256):
        proofs, coms = wl.tile_batch(fx, nb, first=3)
        pb = bytearray(proofs)
        pb[(nb // 2) * fx.proof_len + 128] ^= 1
        proofs = bytes(pb)
        rng = hashlib.shake_256(b"narrow-w%d" % nb).digest(64 * nb)
        res = [c.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng, want_msm=True) for c in contexts]
        for i, r in enumerate(res[1:]):
            assert r == res[0], (nb, FORMS[i + 1])
        assert [i for i in range(nb) if res[0][0][i]] == [nb // 2]
        g = oracle.Gens(64, 1)
        _, ev, em = oracle.verify_batch(g, proofs[:8 * fx.proof_len], coms[:8 * 32], fx.m, fx.n, fx.label, rng[:8 * 64], threads=4) if nb >= 8 else (0, None, None)
        if ev is not None:
            assert res[-2][0][:8] == ev and res[-2][1][:8 * 32] == em
    n, m, nb = 32, 2, 9
    pl = oracle.proof_len(n, m)
    states, proofs, coms = [], b"", b""
    for i in range(nb):
        st = oracle.transcript_append_message(oracle.transcript_new(b"narrow app"), b"session", hashlib.shake_256(b"s%d" % i).digest(24))
        vals = [int.from_bytes(hashlib.shake_256(b"nv%d-%d" % (i, j)).digest(4), "little") for j in range(m)]
        bl = b"".join(hashlib.shake_256(b"nb%d-%d" % (i, j)).digest(31) + b"\x00" for j in range(m))
        pr, cm, _ = oracle.prove_ts(oracle_gens_64_8, vals, bl, n, st, b"seed%d" % i)
        states.append(st)
        proofs += pr
        coms += cm
    pb = bytearray(proofs)
    pb[4 * pl + 128] ^= 1
    proofs = bytes(pb)
    rng = hashlib.shake_256(b"narrow-ts").digest(64 * nb)
    res = [c.rangeproof_verify_batch_ts(n, m, proofs, pl, coms, b"".join(states), rng, want_msm=True, want_transcripts=True) for c in contexts]
    for i, r in enumerate(res[1:]):
        assert r == res[0], FORMS[i + 1]
    assert list(res[0][0]) == [0, 0, 0, 0, 1, 0, 0, 0, 0]
    for i in range(nb):   # the advanced transcripts are the oracle's
        rc, _, ts = oracle.verify_ts(oracle_gens_64_8, proofs[pl * i:pl * (i + 1)], coms[32 * m * i:32 * m * (i + 1)], n, states[i], rng[64 * i:64 * i + 64])
        assert rc == res[0][0][i] and ts == res[0][2][208 * i:208 * (i + 1)]


@pytest.mark.parametrize("fixture,party", [("cfg3_n64_m16", 16), ("cfg4_n64_m32", 32)])
def test_aggregated_shapes_in_narrow_chains(oracle, fixture, party):
    """BASELINE configs 3 and 4 (m = 16 / 32: U = 40 / 58 per-proof points, k = 10 / 11 rounds) as NARROW chains -- a service's single
    aggregated proof per call -- in round 5's form, the rebuilt form without and with second tables: all equal, and equal to the oracle."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture(fixture)
    forms = [FORMS[0], {"narrow_hi_max": 0}, {"narrow_hi_max": 256}, {"narrow_hi_max": 256, "narrow_fused_finish": 0, "narrow_chunk": 5}, {"narrow_hi4_max": 256}]
    ctxs = []
    for f in forms:
        c = bp.Context(0)
        c.set_option("fixed_window_bits", 12)    # (small tables: this test is about the chain's form)
        for k, v in f.items():
            c.set_option(k, v)
        c.gens_create(64, party)
        ctxs.append(c)
    g = oracle.Gens(64, party)
    try:
        for nb in (1, 3, 9):
            proofs, coms = wl.tile_batch(fx, nb, first=1)
            pb = bytearray(proofs)
            if nb > 1:
                pb[(nb - 1) * fx.proof_len + 128] ^= 1     # the last one fails its check
            proofs = bytes(pb)
            rng = hashlib.shake_256(b"narrow-agg-%d-%d" % (party, nb)).digest(64 * nb)
            res = [c.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng, want_msm=True) for c in ctxs]
            for i, r in enumerate(res[1:]):
                assert r == res[0], (fixture, nb, forms[i + 1])
            _, ev, em = oracle.verify_batch(g, proofs, coms, fx.m, fx.n, fx.label, rng, threads=4)
            assert res[0][0] == ev and res[0][1] == em
            assert list(ev) == [0] * (nb - 1) + [1 if nb > 1 else 0]
            for c in ctxs:   # verdicts only
                v = c.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng)
                assert bytes(v) == bytes(ev)
    finally:
        for c in ctxs:
            c.close()

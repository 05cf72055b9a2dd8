"""GPU parity tests of the end-to-end entry point bpgpu_rangeproof_verify_batch
(= RangeProof::from_bytes + verify_multiple_with_rng, src/range_proof/mod.rs:345-452,
504-538) against the CPU oracle: same proofs, same injected batching challenge
=> identical verdicts AND identical 32-byte encodings of the mega-check MSM."""
import hashlib
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx64x8():
    import bulletproofs_amd as bp
    c = bp.Context(0)
    c.gens_create(64, 8)
    yield c
    c.close()


def test_gens_create_matches_reference_derivation(ctx64x8, oracle_gens_64_8):
    """BulletproofGens::new(64, 8) + PedersenGens::default() derived on the device
    (generators.rs:44-53, 157-204) == oracle (itself pinned on Appendix-A KATs)."""
    assert ctx64x8.gens_export() == oracle_gens_64_8.export()
    G, H, B, Bb = ctx64x8.gens_export()
    assert B.hex() == "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"
    assert Bb.hex() == "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"


def test_golden_proofs_verify_end_to_end(ctx64x8, oracle, oracle_gens_64_8, golden):
    """tests/range_proof.rs:16-95 (`deserialize_and_verify`) on the GPU, each case batched with a
    tampered copy and a commitment-swapped copy; verdicts and MSM encodings equal the oracle's."""
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        pr = bytes.fromhex(case["proof"])
        bad = bytearray(pr)
        bad[128] ^= 1
        vc = golden["vc_bytes"]
        swapped = (vc[32:32 * m] + vc[:32]) if m > 1 else vc[32:64]
        proofs = pr + bytes(bad) + pr
        coms = vc[:32 * m] + vc[:32 * m] + swapped
        rng = hashlib.shake_256(b"gold%d-%d" % (n, m)).digest(64 * 3)
        verdict, msm = ctx64x8.rangeproof_verify_batch(n, m, proofs, len(pr), coms, golden["label"], rng, want_msm=True)
        assert list(verdict) == [0, 1, 1], (n, m)
        assert msm[:32] == bytes(32)
        for b in range(3):
            rc, emsm = oracle.verify(oracle_gens_64_8, proofs[len(pr) * b:len(pr) * (b + 1)], coms[32 * m * b:32 * m * (b + 1)], n,
                                     golden["label"], rng[64 * b:64 * b + 64])
            assert verdict[b] == rc and msm[32 * b:32 * b + 32] == emsm, (n, m, b)


def test_golden_with_os_rng(ctx64x8, golden):
    """verify_multiple (thread_rng path): c drawn inside the library."""
    case = golden["cases"][12]   # n = 64, m = 1
    pr = bytes.fromhex(case["proof"])
    v = ctx64x8.rangeproof_verify_batch(64, 1, pr * 5, len(pr), golden["vc_bytes"][:32] * 5, golden["label"], None)
    assert v == bytes(5)


def test_survey_appendix_c_vectors_on_gpu(ctx64x8, golden):
    c = 12345678901234567890123456789
    rng = c.to_bytes(64, "little") * 3
    case = [x for x in golden["cases"] if x["n"] == 8 and x["m"] == 2][0]
    pr = bytes.fromhex(case["proof"])
    bad = bytearray(pr)
    bad[128] ^= 1
    vc = golden["vc_bytes"]
    verdict, msm = ctx64x8.rangeproof_verify_batch(8, 2, pr + bytes(bad) + pr, len(pr), vc[:64] + vc[:64] + vc[32:64] + vc[:32],
                                                   golden["label"], rng, want_msm=True)
    assert list(verdict) == [0, 1, 1]
    assert msm[32:64].hex() == "1830445a3fd1b8fe4b2d9980c062cd3f5ad9fc31236f5d6a3724a28a6856b429"
    assert msm[64:96].hex() == "0a333ed0a6fa3e884490095548f6a8508a2cb44cdea2062e8d0ab36f87d86f21"
    verdict, msm = ctx64x8.rangeproof_verify_batch(8, 2, pr, len(pr), vc[:64], b"other", rng[:64], want_msm=True)
    assert list(verdict) == [1] and msm.hex() == "14bbd613d451719a7e0449b5c8bf67b8b1d0a83e25ada7868417a4df58c40f32"


def test_error_codes_mirror_proof_error(ctx64x8, golden):
    """ProofError mapping (errors.rs:12-54) and check order (from_bytes before verify)."""
    case = golden["cases"][0]   # n = 8, m = 1
    pr = bytes.fromhex(case["proof"])
    vc = golden["vc_bytes"]
    lab = golden["label"]
    rng = hashlib.shake_256(b"err").digest(64 * 8)
    nc = bytearray(pr)
    nc[128:160] = b"\xff" * 32          # t_x not canonical -> FormatError
    nb_ = bytearray(pr)
    nb_[-32:] = b"\xff" * 32            # b not canonical -> FormatError
    ia = bytearray(pr)
    ia[0:32] = bytes(32)                # A = identity encoding -> VerificationError (transcript.rs:75-87)
    us = bytearray(pr)
    us[32] |= 1                         # S does not decode -> VerificationError (mod.rs:445)
    il = bytearray(pr)
    il[224:256] = bytes(32)             # L_0 identity -> VerificationError
    batch = bytes(nc) + bytes(nb_) + bytes(ia) + bytes(us) + bytes(il) + pr
    v = ctx64x8.rangeproof_verify_batch(8, 1, batch, len(pr), vc[:32] * 6, lab, rng[:64 * 6])
    assert list(v) == [2, 2, 1, 1, 1, 0]
    # length-level FormatError: not a multiple of 32 / too short / odd number of ipp elements
    for cut in (len(pr) - 1, 6 * 32, len(pr) - 32):
        v = ctx64x8.rangeproof_verify_batch(8, 1, pr[:cut] * 2, cut, vc[:32] * 2, lab, rng[:128])
        assert list(v) == [2, 2], cut
    # InvalidBitsize, but a malformed proof in the same batch still reports FormatError first
    v = ctx64x8.rangeproof_verify_batch(12, 1, bytes(nc) + pr, len(pr), vc[:32] * 2, lab, rng[:128])
    assert list(v) == [2, 3]
    # n*m != 2^lg(L_vec): VerificationError (ipp.rs:209)
    v = ctx64x8.rangeproof_verify_batch(16, 1, pr, len(pr), vc[:32], lab, rng[:64])
    assert list(v) == [1]
    # too few generators: InvalidGeneratorsLength
    import bulletproofs_amd as bp
    small = bp.Context(0)
    small.gens_create(8, 1)
    assert list(small.rangeproof_verify_batch(8, 1, pr, len(pr), vc[:32], lab, rng[:64])) == [0]
    assert list(small.rangeproof_verify_batch(16, 1, pr, len(pr), vc[:32], lab, rng[:64])) == [4]
    p2 = bytes.fromhex(golden["cases"][1]["proof"])
    assert list(small.rangeproof_verify_batch(8, 2, p2, len(p2), vc[:64], lab, rng[:64])) == [4]
    small.close()
    # empty batch
    assert ctx64x8.rangeproof_verify_batch(8, 1, b"", len(pr), b"", lab, b"") == b""


def test_batch_of_synthetic_64bit_proofs_matches_oracle(ctx64x8, oracle, oracle_gens_64_8):
    """BASELINE config 2 shape (n = 64, m = 1), 200 distinct proofs made by the oracle prover,
    some corrupted in different fields; every verdict and MSM encoding equals the oracle's."""
    nb, n, m = 200, 64, 1
    vals = [int.from_bytes(hashlib.shake_256(b"v%d" % i).digest(8), "little") for i in range(nb)]
    bl = b"".join(hashlib.shake_256(b"b%d" % i).digest(31) + b"\x00" for i in range(nb))
    proofs, coms = oracle.prove_batch(oracle_gens_64_8, vals, bl, m, n, b"cfg2", b"seed", threads=os.cpu_count() or 1)
    pl = oracle.proof_len(n, m)
    pb = bytearray(proofs)
    for i in range(0, nb, 7):            # flip one byte somewhere different in every 7th proof
        pb[i * pl + (i * 37) % pl] ^= 0x40
    proofs = bytes(pb)
    rng = hashlib.shake_256(b"rng-cfg2").digest(64 * nb)
    verdict, msm = ctx64x8.rangeproof_verify_batch(n, m, proofs, pl, coms, b"cfg2", rng, want_msm=True)
    secs, ev, em = oracle.verify_batch(oracle_gens_64_8, proofs, coms, m, n, b"cfg2", rng, threads=os.cpu_count() or 1)
    assert verdict == ev
    assert sum(1 for x in verdict if x == 0) >= nb - nb // 7 - 1
    for b in range(nb):
        if ev[b] in (0, 1) and em[32 * b:32 * b + 32] != b"\xff" * 32:   # oracle writes ff.. when a point fails to decode
            assert msm[32 * b:32 * b + 32] == em[32 * b:32 * b + 32], b


def test_aggregated_m16_matches_oracle(oracle):
    """BASELINE config 3 shape (n = 64, m = 16; N = 2090 terms)."""
    import bulletproofs_amd as bp
    c = bp.Context(0)
    c.gens_create(64, 16)
    g = oracle.Gens(64, 16)
    assert c.gens_export() == g.export()
    nb, n, m = 3, 64, 16
    vals = [(i * 0x9E3779B97F4A7C15) % (1 << 64) for i in range(nb * m)]
    bl = b"".join(hashlib.shake_256(b"bb%d" % i).digest(31) + b"\x00" for i in range(nb * m))
    proofs, coms = oracle.prove_batch(g, vals, bl, m, n, b"agg", b"seed16", threads=os.cpu_count() or 1)
    pl = oracle.proof_len(n, m)
    assert pl == 928
    pb = bytearray(proofs)
    pb[pl + 200] ^= 1
    rng = hashlib.shake_256(b"rng16").digest(64 * nb)
    verdict, msm = c.rangeproof_verify_batch(n, m, bytes(pb), pl, coms, b"agg", rng, want_msm=True)
    secs, ev, em = oracle.verify_batch(g, bytes(pb), coms, m, n, b"agg", rng, threads=3)
    assert list(verdict) == [0, 1, 0] and verdict == ev and msm == em
    c.close()


def test_aggregated_m32_config4_shape(oracle):
    """BASELINE config 4 shape (n = 64, m = 32; N = 4156 terms, proof 992 B), fixture proofs + one corruption."""
    import bulletproofs_amd as bp
    from bulletproofs_amd.workload import load_fixture
    fx = load_fixture("cfg4_n64_m32")
    c = bp.Context(0)
    c.gens_create(64, 32)
    g = oracle.Gens(64, 32)
    nb = 4
    proofs = bytearray(fx.proofs[:nb * fx.proof_len])
    proofs[2 * fx.proof_len + 500] ^= 0x10
    coms = fx.commitments[:nb * 32 * 32]
    rng = hashlib.shake_256(b"rng32").digest(64 * nb)
    verdict, msm = c.rangeproof_verify_batch(64, 32, bytes(proofs), fx.proof_len, coms, fx.label, rng, want_msm=True)
    secs, ev, em = oracle.verify_batch(g, bytes(proofs), coms, 32, 64, fx.label, rng, threads=4)
    assert fx.proof_len == 992 and list(verdict) == [0, 0, 1, 0] and verdict == ev and msm == em
    c.close()


def test_differential_fuzz_against_oracle(ctx64x8, oracle, oracle_gens_64_8, golden):
    """Seeded random single-bit / byte mutations anywhere in the proof or the commitments, several shapes:
    every verdict (incl. FormatError vs VerificationError) and every decodable mega-check encoding must equal
    the oracle's.  Covers non-canonical scalars, undecodable and identity points, wrong challenges."""
    import random
    rnd = random.Random(20260923)
    for case in (golden["cases"][0], golden["cases"][5], golden["cases"][10], golden["cases"][12]):
        n, m = case["n"], case["m"]
        pr = bytes.fromhex(case["proof"])
        pl = len(pr)
        nb = 96
        proofs, coms = bytearray(), bytearray()
        for i in range(nb):
            p, c = bytearray(pr), bytearray(golden["vc_bytes"][:32 * m])
            kind = i % 6
            if kind == 1:
                p[rnd.randrange(pl)] ^= 1 << rnd.randrange(8)
            elif kind == 2:
                c[rnd.randrange(len(c))] ^= 1 << rnd.randrange(8)
            elif kind == 3:
                off = 32 * rnd.randrange(pl // 32)
                p[off:off + 32] = bytes(rnd.randrange(256) for _ in range(32))
            elif kind == 4:
                off = 32 * rnd.randrange(pl // 32)
                p[off:off + 32] = bytes(32)
            elif kind == 5:
                p[32 * rnd.randrange(pl // 32) + 31] |= 0x80
            proofs += p
            coms += c
        rng = hashlib.shake_256(b"fuzz%d-%d" % (n, m)).digest(64 * nb)
        verdict, msm = ctx64x8.rangeproof_verify_batch(n, m, bytes(proofs), pl, bytes(coms), golden["label"], rng, want_msm=True)
        _, ev, em = oracle.verify_batch(oracle_gens_64_8, bytes(proofs), bytes(coms), m, n, golden["label"], rng, threads=os.cpu_count() or 1)
        assert verdict == ev, (n, m, [i for i in range(nb) if verdict[i] != ev[i]][:5])
        assert {0, 1, 2} <= set(verdict)
        for b in range(nb):
            if em[32 * b:32 * b + 32] != b"\xff" * 32 and ev[b] != 2:
                assert msm[32 * b:32 * b + 32] == em[32 * b:32 * b + 32], (n, m, b)


def test_zero_commitments_and_odd_batches(ctx64x8, oracle, oracle_gens_64_8, golden):
    pr = bytes.fromhex(golden["cases"][0]["proof"])
    rng = hashlib.shake_256(b"odd").digest(64 * 67)
    # m = 0: n*m = 0 is not 2^lg(L_vec) -> VerificationError for every well-formed proof (ipp.rs:209)
    assert list(ctx64x8.rangeproof_verify_batch(8, 0, pr * 3, len(pr), b"", golden["label"], rng[:192])) == [1, 1, 1]
    # m = 3 is not a power of two -> VerificationError
    assert list(ctx64x8.rangeproof_verify_batch(8, 3, pr, len(pr), golden["vc_bytes"][:96], golden["label"], rng[:64])) == [1]
    # batch sizes around the wavefront width
    for nb in (1, 63, 64, 65, 67):
        v = ctx64x8.rangeproof_verify_batch(8, 1, pr * nb, len(pr), golden["vc_bytes"][:32] * nb, golden["label"], rng[:64 * nb])
        assert v == bytes(nb), nb
    # empty transcript label
    rc, _ = oracle.verify(oracle_gens_64_8, pr, golden["vc_bytes"][:32], 8, b"", rng[:64])
    assert list(ctx64x8.rangeproof_verify_batch(8, 1, pr, len(pr), golden["vc_bytes"][:32], b"", rng[:64])) == [rc] == [1]


def test_tables_shrink_when_hbm_is_short(golden):
    """Automatic window choice: when the preferred table does not fit the free HBM (another tenant on the device),
    gens_create settles for a smaller window instead of failing; results are unchanged."""
    import torch
    import bulletproofs_amd as bp
    free, total = torch.cuda.mem_get_info(0)
    hog = torch.empty(max(0, free - (24 << 30)), dtype=torch.uint8, device="cuda:0")   # leave ~24 GiB
    c = bp.Context(0)
    c.gens_create(64, 1)                          # preferred: W = 19, 61 GB
    w = c.get_option("fixed_window_bits")
    assert 8 <= w < 19 and c.get_option("fixed_table_bytes") < (24 << 30)
    case = golden["cases"][12]                    # n = 64, m = 1
    pr = bytes.fromhex(case["proof"])
    bad = bytearray(pr)
    bad[128] ^= 1
    v = c.rangeproof_verify_batch(64, 1, pr + bytes(bad), len(pr), golden["vc_bytes"][:32] * 2, golden["label"], None)
    assert list(v) == [0, 1]
    c.close()
    del hog
    torch.cuda.empty_cache()


def test_all_golden_shapes_as_wide_chains(oracle, oracle_gens_64_8, golden):
    """The wide-chain forms (window sums as their own launch, A outside the window sums, one-lane Horner chain on the second stream) on
    every shape of the reference's golden proofs (tests/range_proof.rs:16-95): 2100 copies of each proof with per-copy batching challenges,
    every 7th copy tampered in one of four ways, every 11th with swapped commitments (m > 1).  The oracle verifies the distinct
    (proof variant, challenge) pairs it needs: 64 sampled copies per shape, verdict and mega-check encoding; all verdicts must follow the
    variant's expected verdict."""
    import bulletproofs_amd as bp
    ctx = bp.Context(0, fixed_window_bits=10, horner_lanes=1)   # explicit: the one-lane chain aside on every chain of >= 2048 proofs
    ctx.gens_create(64, 8)
    vc = golden["vc_bytes"]
    nb = 2100
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        pr = bytes.fromhex(case["proof"])
        pl = len(pr)
        variants = [pr]
        for off, mask in ((128, 1), (pl - 64, 2), (3, 4), (224 + 5, 8)):          # t_x, a, the point A, L_0
            b = bytearray(pr)
            b[off] ^= mask
            variants.append(bytes(b))
        swapped = (vc[32:32 * m] + vc[:32]) if m > 1 else vc[:32]
        plist, clist, kinds = [], [], []
        for i in range(nb):
            kind = 1 + (i // 7) % 4 if i % 7 == 6 else 0
            sw = (m > 1 and i % 11 == 10)
            plist.append(variants[kind])
            clist.append(swapped if sw else vc[:32 * m])
            kinds.append((kind, sw))
        proofs, coms = b"".join(plist), b"".join(clist)
        rng = hashlib.shake_256(b"wide-golden-%d-%d" % (n, m)).digest(64 * nb)
        verdict, msm = ctx.rangeproof_verify_batch(n, m, proofs, pl, coms, golden["label"], rng, want_msm=True)
        for i in range(nb):
            assert (verdict[i] == 0) == (kinds[i] == (0, False)), (n, m, i, kinds[i], verdict[i])
        for i in list(range(0, nb, 41)) + [6, 13, 20, 27, 10, nb - 1]:
            rc, emsm = oracle.verify(oracle_gens_64_8, plist[i], clist[i], n, golden["label"], rng[64 * i:64 * i + 64])
            assert verdict[i] == rc, (n, m, i)
            if rc in (0, 1) and emsm != b"\xff" * 32:
                assert msm[32 * i:32 * i + 32] == emsm, (n, m, i)
    ctx.close()

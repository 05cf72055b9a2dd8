"""GPU parity for bpgpu_linear_verify_batch (LinearProof::from_bytes + verify, src/linear_proof.rs:175-236, 240-312,
350-394); sizes of the reference's own tests (linear_proof.rs:470-487: test_helper(n) for n in {1, 16, 32, 64}) plus 2 and
256, against the C oracle (itself pinned by the Python twin: tests/test_oracle.py)."""
import hashlib

import pytest

from test_device_code_on_cpu import _linear_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import bulletproofs_amd as bp
    c = bp.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n", [1, 2, 16, 32, 64])
def test_linear_verify_cases(ctx, oracle, n):
    insts, pl = _linear_cases(oracle, n, b"glin")
    cat = lambda key: b"".join(i[key] for i in insts)
    g0 = insts[0]
    verdict, msm = ctx.linear_verify_batch(n, cat("proof"), pl, cat("C"), g0["G"], g0["F"], g0["B"], cat("b"), label=g0["label"], want_msm=True)
    st = oracle.transcript_new(g0["label"])
    for j, inst in enumerate(insts):
        rc, em = oracle.linear_verify(n, inst["proof"], st, inst["C"], inst["G"], inst["F"], inst["B"], inst["b"])
        assert verdict[j] == rc, (n, j)
        if rc != 2 and j != 3:
            assert msm[32 * j:32 * j + 32] == em, (n, j)          # bit-exact, also for the non-identity results
    assert list(verdict) == [0, 1, 2, 1, 1]
    # wrong n for the proof length (linear_proof.rs:263-265), and a truncated proof (FormatError, :351-360)
    assert list(ctx.linear_verify_batch(2 * n, g0["proof"], pl, g0["C"], g0["G"] * 2, g0["F"], g0["B"], g0["b"] * 2, label=g0["label"])) == [1]
    assert list(ctx.linear_verify_batch(n, g0["proof"][:-1] * 2, pl - 1, g0["C"] * 2, g0["G"], g0["F"], g0["B"], g0["b"] * 2, label=g0["label"])) == [2, 2]


def test_linear_verify_batch_of_many_with_shared_public_vector_and_caller_transcript(ctx, oracle):
    """200 proofs for one public vector b over a transcript that already holds an application message; every 7th proof
    belongs to another transcript and must fail"""
    n, nb = 16, 200
    base = oracle.linear_test_instance(n, b"glin-many")
    st_app = oracle.transcript_append_message(oracle.transcript_new(b"app protocol"), b"ctx", b"session 42")
    st_other = oracle.transcript_new(b"another protocol")
    ell = 2 ** 252 + 27742317777372353535851937790883648493
    proofs, Cs, expect = [], [], []
    for j in range(nb):
        stream = hashlib.shake_256(b"many%d" % j).digest(64 * (n + 1 + 2 * 4 + 2))
        red = lambda i: (int.from_bytes(stream[64 * i:64 * i + 64], "little") % ell).to_bytes(32, "little")
        a = b"".join(red(i) for i in range(n))
        r = red(n)
        c = sum(int.from_bytes(a[32 * i:32 * i + 32], "little") * int.from_bytes(base["b"][32 * i:32 * i + 32], "little") for i in range(n)) % ell
        rcm, Cc = oracle.msm(a + r + c.to_bytes(32, "little"), base["G"] + base["B"] + base["F"])
        assert rcm == 0
        wrong = j % 7 == 6
        rc, pr = oracle.linear_create(n, st_other if wrong else st_app, stream[64 * (n + 1):], Cc, r, a, base["b"], base["G"], base["F"], base["B"])
        assert rc == 0
        proofs.append(pr)
        Cs.append(Cc)
        expect.append(1 if wrong else 0)
    pl = len(proofs[0])
    verdict, msm = ctx.linear_verify_batch(n, b"".join(proofs), pl, b"".join(Cs), base["G"], base["F"], base["B"], base["b"], transcript=st_app,
                                           want_msm=True)
    assert list(verdict) == expect
    for j in (0, 6, 13, 199):
        rc, em = oracle.linear_verify(n, proofs[j], st_app, Cs[j], base["G"], base["F"], base["B"], base["b"])
        assert rc == expect[j] and msm[32 * j:32 * j + 32] == em


def test_linear_verify_n256_uses_the_bucket_msm(ctx, oracle):
    """n = 256: 276 terms per proof; 8 proofs (2208 terms) cross the bucket threshold of the variable-base MSM"""
    n = 256
    insts = [oracle.linear_test_instance(n, b"glin256-%d" % j) for j in range(8)]
    bad = bytearray(insts[5]["proof"])
    bad[-1 - 32] ^= 1
    insts[5] = dict(insts[5], proof=bytes(bad))
    cat = lambda key: b"".join(i[key] for i in insts)
    g0 = insts[0]
    pl = len(g0["proof"])
    verdict, msm = ctx.linear_verify_batch(n, cat("proof"), pl, cat("C"), g0["G"], g0["F"], g0["B"], cat("b"), label=g0["label"], want_msm=True)
    st = oracle.transcript_new(g0["label"])
    for j, inst in enumerate(insts):
        rc, em = oracle.linear_verify(n, inst["proof"], st, inst["C"], inst["G"], inst["F"], inst["B"], inst["b"])
        assert verdict[j] == rc and msm[32 * j:32 * j + 32] == em, j
    assert list(verdict) == [0, 0, 0, 0, 0, 1, 0, 0]


@pytest.mark.parametrize("n", [1, 2, 16, 64])
def test_linear_create_batch_byte_identical(ctx, oracle, n):
    """bpgpu_linear_create_batch == the oracle's LinearProof::create (linear_proof.rs:40-173) byte for byte, on a transcript that
    already holds an application message; the GPU verifier accepts what the GPU prover made and both leave the transcript in
    the same state."""
    nb = 5
    insts = [oracle.linear_test_instance(n, b"glinc-%d-%d" % (n, j)) for j in range(nb)]
    g0 = insts[0]
    st_app = oracle.transcript_append_message(oracle.transcript_new(b"prover app"), b"ctx", b"42")
    cat = lambda key: b"".join(i[key] for i in insts)
    proofs, status, ts_p = ctx.linear_create_batch(n, cat("C"), cat("r"), cat("a"), cat("b"), g0["G"], g0["F"], g0["B"], transcript=st_app,
                                                   rng=cat("rng"), want_transcripts=True)
    pl = len(proofs) // nb
    assert list(status) == [0] * nb and pl == 32 * (2 * (n.bit_length() - 1) + 3)
    for j, inst in enumerate(insts):
        rc, want = oracle.linear_create(n, st_app, inst["rng"], inst["C"], inst["r"], inst["a"], inst["b"], inst["G"], inst["F"], inst["B"])
        assert rc == 0 and proofs[pl * j:pl * (j + 1)] == want, (n, j)
    verdict, ts_v = ctx.linear_verify_batch(n, proofs, pl, cat("C"), g0["G"], g0["F"], g0["B"], cat("b"), transcript=st_app, want_transcripts=True)
    assert list(verdict) == [0] * nb and ts_v == ts_p


def test_linear_create_batch_of_many_with_os_randomness(ctx, oracle):
    """300 proofs for one public vector, randomness from the OS CSPRNG: every proof verifies (GPU and, spot-checked, oracle),
    no two proofs share their first L; a non-canonical secret scalar is reported in status instead of proved"""
    n, nb = 16, 300
    base = oracle.linear_test_instance(n, b"glinc-many")
    ell = 2 ** 252 + 27742317777372353535851937790883648493
    As, rs, Cs = [], [], []
    for j in range(nb):
        stream = hashlib.shake_256(b"cmany%d" % j).digest(64 * (n + 1))
        red = lambda i: (int.from_bytes(stream[64 * i:64 * i + 64], "little") % ell).to_bytes(32, "little")
        a = b"".join(red(i) for i in range(n))
        r = red(n)
        c = sum(int.from_bytes(a[32 * i:32 * i + 32], "little") * int.from_bytes(base["b"][32 * i:32 * i + 32], "little") for i in range(n)) % ell
        rcm, Cc = oracle.msm(a + r + c.to_bytes(32, "little"), base["G"] + base["B"] + base["F"])
        assert rcm == 0
        As.append(a)
        rs.append(r)
        Cs.append(Cc)
    As[7] = b"\xff" * 32 + As[7][32:]
    proofs, status = ctx.linear_create_batch(n, b"".join(Cs), b"".join(rs), b"".join(As), base["b"], base["G"], base["F"], base["B"], label=b"many")
    pl = len(proofs) // nb
    assert [j for j in range(nb) if status[j] != 0] == [7]
    verdict = ctx.linear_verify_batch(n, proofs, pl, b"".join(Cs), base["G"], base["F"], base["B"], base["b"], label=b"many")
    assert [j for j in range(nb) if verdict[j] != 0] == [7]
    assert len({proofs[pl * j:pl * j + 32] for j in range(nb) if j != 7}) == nb - 1
    st = oracle.transcript_new(b"many")
    for j in (0, 150, 299):
        assert oracle.linear_verify(n, proofs[pl * j:pl * (j + 1)], st, Cs[j], base["G"], base["F"], base["B"], base["b"])[0] == 0


@pytest.mark.parametrize("n", [1, 16, 64])
def test_linear_verify_with_the_contexts_generators(oracle, n):
    """G = F = B = NULL: the bases are the context's bp_gens.share(0).G(n), pc_gens.B, pc_gens.B_blinding (the reference's own
    test shape, linear_proof.rs:405-411) and their coefficients go through the fixed-base window tables; verdicts, results and
    returned transcripts equal the explicit-bases path and the oracle."""
    import bulletproofs_amd as bp
    c = bp.Context(0)
    c.gens_create(64, 2)                         # capacity 64 >= n, two parties: only party 0's G are used
    insts, pl = _linear_cases(oracle, n, b"gfix")
    cat = lambda key: b"".join(i[key] for i in insts)
    g0 = insts[0]
    G_all, _, Bp, Bb = c.gens_export()
    assert G_all[:32 * n] == g0["G"] and Bp == g0["F"] and Bb == g0["B"]
    v_fix, m_fix, t_fix = c.linear_verify_batch(n, cat("proof"), pl, cat("C"), None, None, None, cat("b"), label=g0["label"], want_msm=True,
                                                want_transcripts=True)
    v_gen, m_gen, t_gen = c.linear_verify_batch(n, cat("proof"), pl, cat("C"), g0["G"], g0["F"], g0["B"], cat("b"), label=g0["label"], want_msm=True,
                                                want_transcripts=True)
    assert v_fix == v_gen and list(v_fix) == [0, 1, 2, 1, 1] and t_fix == t_gen
    st = oracle.transcript_new(g0["label"])
    for j in (0, 1, 4):
        rc, em = oracle.linear_verify(n, insts[j]["proof"], st, insts[j]["C"], g0["G"], g0["F"], g0["B"], insts[j]["b"])
        assert v_fix[j] == rc and m_fix[32 * j:32 * j + 32] == em == m_gen[32 * j:32 * j + 32], (n, j)
    # the prover over the same bases: byte-identical to the oracle's (and so to the explicit-bases path) through the window tables
    if True:
        rc, want = oracle.linear_create(n, st, g0["rng"], g0["C"], g0["r"], g0["a"], g0["b"], g0["G"], g0["F"], g0["B"])
        made, stat = c.linear_create_batch(n, g0["C"] * 3, g0["r"] * 3, g0["a"] * 3, g0["b"], None, None, None, label=g0["label"], rng=g0["rng"] * 3)
        assert rc == 0 and stat == bytes(3) and made == want * 3
    # wrong n for the proof length -> VerificationError without touching the tables; more generators than loaded -> an error, not a verdict
    assert list(c.linear_verify_batch(2 * n, g0["proof"], pl, g0["C"], None, None, None, g0["b"] * 2, label=g0["label"])) == [1]
    big = oracle.linear_test_instance(128, b"gfix-big")
    with pytest.raises(bp.BpgpuError):
        c.linear_verify_batch(128, big["proof"], len(big["proof"]), big["C"], None, None, None, big["b"], label=big["label"])
    c.close()


@pytest.mark.parametrize("n,fixed", [(4, False), (16, True), (64, False)])
def test_linear_differential_fuzz_against_oracle(oracle, n, fixed):
    """Seeded single-bit / 32-byte mutations anywhere in the proof, the commitment or the public vector: every verdict (incl.
    FormatError vs VerificationError) and every computed result equals the oracle's -- explicit bases and generator-table mode."""
    import random
    import bulletproofs_amd as bp
    rnd = random.Random(20260924 + n)
    c = bp.Context(0)
    c.gens_create(64, 1)
    base = oracle.linear_test_instance(n, b"glin-fuzz-%d" % n)
    pl, nb = len(base["proof"]), 120
    proofs, Cs, bs = bytearray(), bytearray(), bytearray()
    for i in range(nb):
        p, cc, b = bytearray(base["proof"]), bytearray(base["C"]), bytearray(base["b"])
        kind = i % 8
        if kind == 1:
            p[rnd.randrange(pl)] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            cc[rnd.randrange(32)] ^= 1 << rnd.randrange(8)
        elif kind == 3:
            off = 32 * rnd.randrange(pl // 32)
            p[off:off + 32] = bytes(rnd.randrange(256) for _ in range(32))
        elif kind == 4:
            off = 32 * rnd.randrange(pl // 32)
            p[off:off + 32] = bytes(32)
        elif kind == 5:
            p[32 * rnd.randrange(pl // 32) + 31] |= 0x80
        elif kind == 6:
            b[32 * rnd.randrange(n) + rnd.randrange(31)] ^= 1 << rnd.randrange(8)          # stays canonical (top byte untouched)
        elif kind == 7:
            p[pl - 64 + rnd.randrange(64)] ^= 1 << rnd.randrange(8)                        # a or r
        proofs += p
        Cs += cc
        bs += b
    args = (None, None, None) if fixed else (base["G"], base["F"], base["B"])
    verdict, msm = c.linear_verify_batch(n, bytes(proofs), pl, bytes(Cs), *args, bytes(bs), label=base["label"], want_msm=True)
    st = oracle.transcript_new(base["label"])
    seen = set()
    for i in range(nb):
        rc, em = oracle.linear_verify(n, bytes(proofs[pl * i:pl * (i + 1)]), st, bytes(Cs[32 * i:32 * i + 32]), base["G"], base["F"], base["B"],
                                      bytes(bs[32 * n * i:32 * n * (i + 1)]))
        assert verdict[i] == rc, (n, i, i % 8, verdict[i], rc)
        seen.add(rc)
        if rc != 2 and em not in (b"\xff" * 32, bytes(32)):
            assert msm[32 * i:32 * i + 32] == em, (n, i, i % 8)
    assert seen == {0, 1, 2}
    c.close()

"""bpgpu_ipp_create_batch: InnerProductProof::create (src/inner_product_proof.rs:38-193) for a batch of proofs, all
multiscalar multiplications on the GPU engine (csrc/ipp_prover.h).  Parity: proofs byte-identical to the oracle's
restatement of the reference algorithm, for the sizes the range proofs use (n*m = 64 and 1024) and the reference's own
test sizes; the GPU verifier accepts them; error statuses."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu
L = 2**252 + 27742317777372353535851937790883648493


def _sc(tag):
    return (int.from_bytes(hashlib.shake_256(tag).digest(64), "little") % L).to_bytes(32, "little")


@pytest.fixture(scope="module")
def ctx():
    import bulletproofs_amd as bp
    c = bp.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,nb", [(1, 3), (2, 5), (4, 5), (32, 9), (64, 40), (1024, 6)])
def test_batched_create_is_byte_identical_to_oracle(ctx, oracle, n, nb):
    insts = [oracle.ipp_test_instance(n, b"innerproducttest", b"gpv-%d-%d" % (n, j % 3)) for j in range(nb)]
    a = [b"".join(_sc(b"ga%d-%d-%d" % (n, j, i)) for i in range(n)) for j in range(nb)]
    b = [b"".join(_sc(b"gb%d-%d-%d" % (n, j, i)) for i in range(n)) for j in range(nb)]
    cat = lambda k: b"".join(x[k] for x in insts)
    pl = 32 * (2 * (n.bit_length() - 1) + 2)
    proofs, st = ctx.ipp_create_batch(n, cat("Q"), cat("Gf"), cat("Hf"), cat("G"), cat("H"), b"".join(a), b"".join(b), label=b"innerproducttest")
    assert st == bytes(nb)
    for j in range(nb if n < 1024 else 2):
        rc, exp = oracle.ipp_create(n, b"innerproducttest", insts[j]["Q"], insts[j]["Hf"], insts[j]["G"], insts[j]["H"], a[j], b[j])
        assert rc == 0 and proofs[pl * j:pl * (j + 1)] == exp, (n, j)
    # the verifier (GPU and oracle) accepts them against P = <a,G> + <b, Hf o H> + <a,b> Q
    for j in range(min(nb, 3)):
        sc = b"".join(a[j][32 * i:32 * i + 32] for i in range(n)) + \
             b"".join((int.from_bytes(b[j][32 * i:32 * i + 32], "little") * int.from_bytes(insts[j]["Hf"][32 * i:32 * i + 32], "little") % L).to_bytes(32, "little") for i in range(n)) + \
             (sum(int.from_bytes(a[j][32 * i:32 * i + 32], "little") * int.from_bytes(b[j][32 * i:32 * i + 32], "little") for i in range(n)) % L).to_bytes(32, "little")
        P = oracle.msm(sc, insts[j]["G"] + insts[j]["H"] + insts[j]["Q"])[1]
        v = ctx.ipp_verify_batch(n, proofs[pl * j:pl * (j + 1)], pl, b"innerproducttest", insts[j]["Gf"], insts[j]["Hf"], P, insts[j]["Q"], insts[j]["G"], insts[j]["H"])
        assert list(v) == [0], (n, j)
    # bases shared by the batch + a transcript with history (state handed over instead of a label)
    ts = oracle.transcript_append_message(oracle.transcript_new(b"app"), b"ctx", b"x" * 37)
    proofs2, st2 = ctx.ipp_create_batch(n, cat("Q"), cat("Gf"), cat("Hf"), insts[0]["G"], insts[0]["H"], b"".join(a), b"".join(b), transcript=ts)
    proofs3, st3 = ctx.ipp_create_batch(n, cat("Q"), cat("Gf"), cat("Hf"), insts[0]["G"] * nb, insts[0]["H"] * nb, b"".join(a), b"".join(b), transcript=ts)
    assert st2 == st3 == bytes(nb) and proofs2 == proofs3 and (n == 1 or proofs2 != proofs)


def test_create_error_statuses(ctx, oracle):
    n, nb = 8, 4
    inst = oracle.ipp_test_instance(n, b"innerproducttest", b"err")
    a = b"".join(_sc(b"ea%d" % i) for i in range(n))
    b = b"".join(_sc(b"eb%d" % i) for i in range(n))
    G = bytearray(inst["G"] * nb)
    G[32 * n * 1 + 3 * 32] |= 1                     # proof 1: an undecodable G point
    aa = bytearray(a * nb)
    aa[32 * n * 2:32 * n * 2 + 32] = L.to_bytes(32, "little")   # proof 2: a non-canonical scalar
    proofs, st = ctx.ipp_create_batch(n, inst["Q"] * nb, inst["Gf"] * nb, inst["Hf"] * nb, bytes(G), inst["H"] * nb, bytes(aa), b * nb, label=b"innerproducttest")
    assert list(st) == [0, 1, 2, 0]
    pl = 32 * 8
    rc, exp = oracle.ipp_create(n, b"innerproducttest", inst["Q"], inst["Hf"], inst["G"], inst["H"], a, b)
    assert proofs[:pl] == exp and proofs[3 * pl:] == exp
    import bulletproofs_amd as bp
    with pytest.raises(bp.BpgpuError):
        ctx.ipp_create_batch(6, inst["Q"], inst["Gf"][:192], inst["Hf"][:192], inst["G"][:192], inst["H"][:192], a[:192], b[:192])


@pytest.mark.parametrize("n,m,nb", [(64, 1, 48), (32, 4, 10), (8, 2, 5), (64, 16, 3)])
def test_rangeproof_prove_batch_is_byte_identical_to_oracle(oracle, n, m, nb):
    """bpgpu_rangeproof_prove_batch == the oracle's prove_multiple_with_rng restatement (mod.rs:234-288, party.rs, dealer.rs)
    on the same random scalars: proofs, commitments and final transcripts byte for byte; the GPU verifier accepts the
    proofs on the same (pre-bound) transcript and rejects them on another one."""
    import bulletproofs_amd as bp
    ctx = bp.Context(0, fixed_window_bits=12)
    ctx.gens_create(n, m)
    g = oracle.Gens(n, m)
    vals = [int.from_bytes(hashlib.shake_256(b"gv%d-%d-%d" % (n, m, i)).digest(8), "little") % (1 << n) for i in range(nb * m)]
    vals[0] = (1 << n) - 1
    vals[-1] = 0
    bl = b"".join(hashlib.shake_256(b"gb%d-%d-%d" % (n, m, i)).digest(32) for i in range(nb * m))
    per = 64 * (m * (2 * n + 2) + 2 * m)
    seeds = [b"grpp-%d-%d-%d" % (n, m, p) for p in range(nb)]
    rng = b"".join(hashlib.shake_256(sd).digest(per) for sd in seeds)
    st0 = oracle.transcript_append_message(oracle.transcript_new(b"prover"), b"ctx", b"gpu")
    pl = oracle.proof_len(n, m)
    proofs, coms, ts = ctx.rangeproof_prove_batch(n, m, vals, bl, transcript=st0, rng=rng, want_transcripts=True)
    for p in range(nb if n * m <= 256 else 2):
        epr, ecm, ets = oracle.prove_ts(g, vals[p * m:(p + 1) * m], bl[32 * m * p:32 * m * (p + 1)], n, st0, seeds[p])
        assert coms[32 * m * p:32 * m * (p + 1)] == ecm and proofs[pl * p:pl * (p + 1)] == epr and ts[208 * p:208 * (p + 1)] == ets, (n, m, p)
    v, ts_v = ctx.rangeproof_verify_batch_ts(n, m, proofs, pl, coms, st0, bytes(64 * nb), want_transcripts=True)
    assert v == bytes(nb) and ts_v == ts            # prover and verifier leave the same transcript
    assert all(x == 1 for x in ctx.rangeproof_verify_batch(n, m, proofs, pl, coms, b"prover", bytes(64 * nb)))
    # label form + library randomness: proofs verify, and differ from call to call
    p1, c1 = ctx.rangeproof_prove_batch(n, m, vals, bl, label=b"lbl")
    p2, c2 = ctx.rangeproof_prove_batch(n, m, vals, bl, label=b"lbl")
    assert c1 == c2 == coms and p1 != p2
    assert ctx.rangeproof_verify_batch(n, m, p1, pl, c1, b"lbl") == bytes(nb)
    # parameter errors of the reference: InvalidBitsize, InvalidAggregation, InvalidGeneratorsLength, value out of range
    for bad_n, bad_m, bad_vals in ((24, m, vals), (n, 3, vals[:3 * (len(vals) // 3) if len(vals) >= 3 else 0] or [0, 0, 0]), (n, 2 * m, vals * 2)):
        with pytest.raises(bp.BpgpuError):
            ctx.rangeproof_prove_batch(bad_n, bad_m, bad_vals, bytes(32 * len(bad_vals)), label=b"x")
    if n < 64:
        with pytest.raises(bp.BpgpuError):
            ctx.rangeproof_prove_batch(n, m, [1 << n] + vals[1:], bl, label=b"x")
    ctx.close()


def test_reference_api_prove_then_verify(oracle):
    from bulletproofs_amd import BulletproofGens, RangeProof, Transcript
    bp_gens = BulletproofGens(64, 4, fixed_window_bits=12)
    pc_gens = bp_gens.pedersen()
    tp = Transcript(b"doctest example")
    secrets = [4242344947, 3718732727, 2255562556, 2526146994]           # README.md of the reference: the aggregated example
    blindings = [hashlib.shake_256(b"rb%d" % i).digest(31) + b"\x00" for i in range(4)]
    proof, commitments = RangeProof.prove_multiple_with_rng(bp_gens, pc_gens, tp, secrets, blindings, 32)
    tv = Transcript(b"doctest example")
    assert RangeProof.from_bytes(proof.to_bytes()).verify_multiple(bp_gens, pc_gens, tv, commitments, 32) is None
    assert tv.state == tp.state
    g = oracle.Gens(64, 4)
    assert oracle.verify(g, proof.to_bytes(), b"".join(commitments), 32, b"doctest example", bytes(64))[0] == 0


@pytest.fixture(scope="module")
def gens64x8():
    from bulletproofs_amd import BulletproofGens
    g = BulletproofGens(64, 8, fixed_window_bits=12)
    yield g
    g.ctx.close()


@pytest.mark.parametrize("n,m", [(32, 1), (32, 2), (32, 4), (32, 8), (64, 1), (64, 2), (64, 4), (64, 8)])
def test_singleparty_create_and_verify_helper(gens64x8, oracle, n, m):
    """src/range_proof/mod.rs:633-725: create_and_verify_n_{32,64}_m_{1,2,4,8} -- prover's scope (prove, serialize), verifier's
    scope (deserialize, verify) -- through the Python mirror, both scopes on the GPU; the oracle verifies the same bytes."""
    from bulletproofs_amd import RangeProof, Transcript, VerificationError
    max_bitsize, max_parties = 64, 8
    bp_gens, pc_gens = gens64x8, gens64x8.pedersen()
    assert (bp_gens.gens_capacity, bp_gens.party_capacity) == (max_bitsize, max_parties)
    # 1. prover's scope
    rnd = hashlib.shake_256(b"create_and_verify %d %d" % (n, m)).digest(8 * m + 64 * m)
    values = [int.from_bytes(rnd[8 * i:8 * i + 8], "little") % (1 << n) for i in range(m)]       # rng.gen_range(min, max)
    ell = 2 ** 252 + 27742317777372353535851937790883648493
    blindings = [(int.from_bytes(rnd[8 * m + 64 * i:8 * m + 64 * i + 64], "little") % ell).to_bytes(32, "little") for i in range(m)]
    tp = Transcript(b"AggregatedRangeProofTest")
    proof, value_commitments = RangeProof.prove_multiple_with_rng(bp_gens, pc_gens, tp, values, blindings, n)
    proof_bytes = proof.to_bytes()
    assert len(proof_bytes) == 32 * (9 + 2 * ((n * m).bit_length() - 1)) and len(value_commitments) == m
    # 2. verifier's scope
    parsed = RangeProof.from_bytes(proof_bytes)
    tv = Transcript(b"AggregatedRangeProofTest")
    assert parsed.verify_multiple(bp_gens, pc_gens, tv, value_commitments, n) is None
    assert tv.state == tp.state
    g = oracle.Gens(max_bitsize, max_parties)
    assert oracle.verify(g, proof_bytes, b"".join(value_commitments), n, b"AggregatedRangeProofTest", bytes(64))[0] == 0
    # a commitment from another proof makes it fail (the aggregation tests' negative direction)
    if m > 1:
        swapped = [value_commitments[1], value_commitments[0]] + value_commitments[2:]
        with pytest.raises(VerificationError):
            parsed.verify_multiple(bp_gens, pc_gens, Transcript(b"AggregatedRangeProofTest"), swapped, n)


def test_prover_entry_points_leave_no_secrets_in_the_staging_buffers(oracle):
    """The reference zeroizes its parties' secrets on Drop (party.rs:148-260); the batched provers clear what they staged --
    pinned block, device IO buffer, working sets, MSM arena -- before they return, on success and on the error paths
    (`staging_residue`: non-zero bytes left in those buffers)."""
    import hashlib
    import bulletproofs_amd as bp
    ctx = bp.Context(0, fixed_window_bits=12)
    ctx.gens_create(64, 2)
    nb, n, m = 33, 64, 2
    vals = [int.from_bytes(hashlib.shake_256(b"zv%d" % i).digest(8), "little") for i in range(nb * m)]
    bl = b"".join(hashlib.shake_256(b"zb%d" % i).digest(31) + b"\x00" for i in range(nb * m))
    proofs, coms = ctx.rangeproof_prove_batch(n, m, vals, bl, label=b"zeroize")
    assert ctx.get_option("staging_residue") == 0
    assert ctx.rangeproof_verify_batch(n, m, proofs, len(proofs) // nb, coms, b"zeroize") == bytes(nb)
    inst = oracle.linear_test_instance(16, b"zeroize-linear")
    made, status = ctx.linear_create_batch(16, inst["C"], inst["r"], inst["a"], inst["b"], None, None, None, label=inst["label"], rng=inst["rng"])
    assert status == bytes(1) and made == inst["proof"] and ctx.get_option("staging_residue") == 0
    ip = oracle.ipp_test_instance(8, b"zeroize-ipp", b"s")
    ctx.close()


def test_increase_capacity_keeps_custom_pedersen_bases(oracle):
    """BulletproofGens::increase_capacity (generators.rs:177-204) on a context whose Pedersen bases were loaded by the caller: the
    rebuilt tables keep those bases (they used to revert to the defaults)."""
    import bulletproofs_amd as bp
    g = bp.BulletproofGens(8, 1)
    G, H, B, Bb = g.ctx.gens_export()
    g.ctx.gens_load(8, 1, G, H, Bb, B)                      # custom bases: the default pair swapped
    g.increase_capacity(16)
    G2, H2, B2, Bb2 = g.ctx.gens_export()
    assert (B2, Bb2) == (Bb, B) and G2[:32 * 8] == G and len(G2) == 32 * 16
    og = oracle.Gens(16, 1).export()
    assert G2 == og[0] and H2 == og[1]

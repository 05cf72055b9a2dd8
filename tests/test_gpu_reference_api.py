"""The reference's own integration test, tests/range_proof.rs::deserialize_and_verify (lines 16-95),
written against the host-side mirrors of the crate API: Python (bulletproofs_amd.api) and C++
(include/bulletproofs.hpp, compiled here with g++ and linked against libbpgpu.so)."""
import hashlib
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_deserialize_and_verify(golden):
    from bulletproofs_amd import BulletproofGens, RangeProof, Transcript, VerificationError, InvalidBitsize, FormatError
    # proofs[i][j] has bitsize n = 8 << i, aggregation size m = 1 << j
    proofs = [[golden["cases"][4 * i + j]["proof"] for j in range(4)] for i in range(4)]
    vc = [bytes.fromhex(v) for v in golden["value_commitments"]]
    bp_gens = BulletproofGens(64, 8)
    pc_gens = bp_gens.pedersen()
    for i in range(4):
        for j in range(4):
            n, m = 8 << i, 1 << j
            proof = RangeProof.from_bytes(bytes.fromhex(proofs[i][j]))
            transcript = Transcript(b"Deserialize-And-Verify Test")
            assert proof.verify_multiple(bp_gens, pc_gens, transcript, vc[0:m], n) is None   # == Ok(())
            with pytest.raises(VerificationError):
                proof.verify_multiple(bp_gens, pc_gens, Transcript(b"another label"), vc[0:m], n)
            before = transcript.state
            with pytest.raises(InvalidBitsize):
                proof.verify_multiple(bp_gens, pc_gens, transcript, vc[0:m], 24)
            assert transcript.state == before      # the bitsize check precedes any transcript operation (mod.rs:358-360)
    with pytest.raises(FormatError):
        RangeProof.from_bytes(bytes.fromhex(proofs[0][0])[:-1])
    res = RangeProof.verify_batch(bp_gens, pc_gens, Transcript(b"Deserialize-And-Verify Test"),
                                  [bytes.fromhex(proofs[3][0])] * 3, [[vc[0]], [vc[1]], [vc[0]]], 64)
    assert res[0] is None and res[1] == VerificationError() and res[2] is None
    # the batch-combined entry point returns the same verdicts (clean batch: one identity check; otherwise the fallback)
    clean = RangeProof.verify_batch_combined(bp_gens, pc_gens, Transcript(b"Deserialize-And-Verify Test"),
                                             [bytes.fromhex(proofs[3][0])] * 4, [[vc[0]]] * 4, 64)
    assert clean == [None] * 4
    res2 = RangeProof.verify_batch_combined(bp_gens, pc_gens, Transcript(b"Deserialize-And-Verify Test"),
                                            [bytes.fromhex(proofs[3][0])] * 3, [[vc[0]], [vc[1]], [vc[0]]], 64)
    assert res2[0] is None and res2[1] == VerificationError() and res2[2] is None


def test_cpp_mirror_deserialize_and_verify(golden, tmp_path):
    inc = tmp_path / "golden_vectors.inc"
    with open(inc, "w") as f:
        f.write("static const char *GOLDEN_PROOFS[4][4] = {\n")
        for i in range(4):
            f.write("  {" + ", ".join('"%s"' % golden["cases"][4 * i + j]["proof"] for j in range(4)) + "},\n")
        f.write("};\nstatic const char *GOLDEN_VC[8] = {" + ", ".join('"%s"' % v for v in golden["value_commitments"]) + "};\n")
        f.write('static const char *GOLDEN_LABEL = "%s";\n' % golden["transcript_label"])
    exe = tmp_path / "range_proof_test"
    libdir = os.path.join(ROOT, "bulletproofs_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
                           os.path.join(ROOT, "tests", "cpp", "range_proof_test.cpp"), "-o", str(exe),
                           "-L", libdir, "-lbpgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "deserialize_and_verify: ok" in out.stdout


def _linear_replay(oracle, inst):
    """the transcript LinearProof::verify leaves behind (linear_proof.rs:196-208), replayed with the oracle's Merlin"""
    n = inst["n"]
    st = oracle.transcript_new(inst["label"])
    st = oracle.transcript_append_message(st, b"dom-sep", b"ipp v1")
    st = oracle.transcript_append_message(st, b"n", n.to_bytes(8, "little"))
    st = oracle.transcript_append_message(st, b"C", inst["C"])
    for i in range(n):
        st = oracle.transcript_append_message(st, b"b_i", inst["b"][32 * i:32 * i + 32])
    for i in range(n):
        st = oracle.transcript_append_message(st, b"G_i", inst["G"][32 * i:32 * i + 32])
    st = oracle.transcript_append_message(st, b"F", inst["F"])
    st = oracle.transcript_append_message(st, b"B", inst["B"])
    k = n.bit_length() - 1
    for j in range(k):
        st = oracle.transcript_append_message(st, b"L", inst["proof"][64 * j:64 * j + 32])
        st = oracle.transcript_append_message(st, b"R", inst["proof"][64 * j + 32:64 * j + 64])
        st, _ = oracle.transcript_challenge_bytes(st, b"x_j", 64)
    st = oracle.transcript_append_message(st, b"S", inst["proof"][64 * k:64 * k + 32])
    st, _ = oracle.transcript_challenge_bytes(st, b"x_star", 64)
    return st


@pytest.mark.parametrize("n", [1, 16, 32, 64])
def test_linear_proof_test_helper(oracle, n):
    """src/linear_proof.rs:401-466 test_helper(n) for the sizes of :470-487, through the Python mirror of LinearProof: the
    proof comes from the oracle's prover, verification from the GPU; the transcript must come back advanced exactly as the
    reference leaves it."""
    from bulletproofs_amd import Context, LinearProof, Transcript, VerificationError, FormatError, InvalidGeneratorsLength
    ctx = Context(0)
    inst = oracle.linear_test_instance(n, b"api-lin-%d" % n)
    sp = lambda bs: [bs[32 * i:32 * i + 32] for i in range(len(bs) // 32)]
    G, b = sp(inst["G"]), sp(inst["b"])
    proof = LinearProof.from_bytes(inst["proof"])
    assert proof.serialized_size() == len(proof.to_bytes()) == 32 * (2 * (n.bit_length() - 1) + 3)
    t = Transcript(b"linearprooftest")
    assert proof.verify(t, inst["C"], G, inst["F"], inst["B"], b, ctx) is None
    assert t.state == _linear_replay(oracle, inst)
    again = LinearProof.from_bytes(proof.to_bytes())
    assert again.verify(Transcript(b"linearprooftest"), inst["C"], G, inst["F"], inst["B"], b, ctx) is None
    # create on the GPU (:424-436): byte-identical to the oracle's given the same rng bytes; prover and verifier transcripts agree
    tp = Transcript(b"linearprooftest")
    made = LinearProof.create(tp, inst["rng"], inst["C"], inst["r"], sp(inst["a"]), b, G, inst["F"], inst["B"], ctx)
    assert made.to_bytes() == inst["proof"] and tp.state == t.state
    fresh = LinearProof.create(Transcript(b"linearprooftest"), None, inst["C"], inst["r"], sp(inst["a"]), b, G, inst["F"], inst["B"], ctx)
    assert fresh.to_bytes() != inst["proof"] and fresh.verify(Transcript(b"linearprooftest"), inst["C"], G, inst["F"], inst["B"], b, ctx) is None
    if n > 2:
        with pytest.raises(ValueError):                                           # InvalidInputLength (:64-70): 3 is not a power of two
            LinearProof.create(Transcript(b"x"), None, inst["C"], inst["r"], sp(inst["a"])[:3], b[:3], G[:3], inst["F"], inst["B"], ctx)
    with pytest.raises(VerificationError):
        proof.verify(t, inst["C"], G, inst["F"], inst["B"], b, ctx)             # the advanced transcript is another statement
    with pytest.raises(VerificationError):
        proof.verify(Transcript(b"linearprooftest"), inst["B"], G, inst["F"], inst["B"], b, ctx)
    with pytest.raises(InvalidGeneratorsLength):
        proof.verify(Transcript(b"linearprooftest"), inst["C"], G + G[:1], inst["F"], inst["B"], b, ctx)
    with pytest.raises(FormatError):
        LinearProof.from_bytes(inst["proof"][:-32])
    res = LinearProof.verify_batch(ctx, Transcript(b"linearprooftest"), [proof, again, proof], [inst["C"], inst["C"], inst["F"]], G, inst["F"],
                                   inst["B"], [b, b, b])
    assert res[0] is None and res[1] is None and res[2] == VerificationError()
    ctx.close()


def test_cpp_mirror_linear_proof(oracle, tmp_path):
    cases = [oracle.linear_test_instance(n, b"cpp-lin-%d-%d" % (n, j)) for n, j in ((1, 0), (16, 0), (16, 1), (16, 2), (32, 0), (64, 0))]
    with open(tmp_path / "linear_vectors.inc", "w") as f:
        f.write("struct linear_case { size_t n; const char *proof, *C, *G, *F, *B, *b, *a, *r, *rng; };\nstatic const linear_case LINEAR_CASES[] = {\n")
        for c in cases:
            f.write('  {%d, %s},\n' % (c["n"], ", ".join('"%s"' % c[key].hex() for key in ("proof", "C", "G", "F", "B", "b", "a", "r", "rng"))))
        f.write("};\n")
    exe = tmp_path / "linear_proof_test"
    libdir = os.path.join(ROOT, "bulletproofs_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
                           os.path.join(ROOT, "tests", "cpp", "linear_proof_test.cpp"), "-o", str(exe),
                           "-L", libdir, "-lbpgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "linear_proof: ok" in out.stdout


def test_generators_tests_of_the_reference(oracle):
    """src/generators.rs:300-356: aggregated_gens_iter_matches_flat_gens and resizing_small_gens_matches_creating_bigger_gens,
    through the Python mirror (generators derived on the GPU), plus equality with the oracle's chain."""
    from bulletproofs_amd import BulletproofGens
    gens = BulletproofGens(64, 8)
    for n, m in ((64, 8), (64, 4), (64, 2), (64, 1), (32, 8), (32, 4), (32, 2), (32, 1), (16, 8), (16, 4), (16, 2), (16, 1)):
        agg_G, agg_H = gens.G(n, m), gens.H(n, m)
        flat_G = [g for j in range(m) for g in gens.share(j).G(n)]
        flat_H = [h for j in range(m) for h in gens.share(j).H(n)]
        assert agg_G == flat_G and agg_H == flat_H and len(agg_G) == n * m
    resized = BulletproofGens(32, 8)
    before = resized.G(32, 8)
    resized.increase_capacity(16)                      # not larger: nothing happens
    assert resized.gens_capacity == 32 and resized.G(32, 8) == before
    resized.increase_capacity(64)
    for n, m in ((64, 8), (32, 8), (16, 8)):
        assert gens.G(n, m) == resized.G(n, m) and gens.H(n, m) == resized.H(n, m)
    oG, oH, oB, oBb = oracle.Gens(64, 8).export()
    assert b"".join(gens.G(64, 8)) == oG and b"".join(gens.H(64, 8)) == oH
    pc = resized.pedersen()
    assert (pc.B, pc.B_blinding) == (oB, oBb)
    # PedersenGens::commit (generators.rs:38-42) against an oracle MSM (the reference-held vc[j] are checked in
    # test_commit_and_prover_commitments_reproduce_the_references_vectors below); then prove_single / verify_single with thread_rng
    from bulletproofs_amd import RangeProof, Transcript
    blind = hashlib.shake_256(b"commit").digest(31) + b"\x00"
    want = oracle.msm((1037578891).to_bytes(32, "little") + blind, oB + oBb)[1]
    assert resized.commit(1037578891, blind) == want
    proof, V = RangeProof.prove_single(resized, pc, Transcript(b"doctest example"), 1037578891, blind, 32)     # README.md:120 of the reference
    assert V == want and proof.verify_single(resized, pc, Transcript(b"doctest example"), V, 32) is None


def test_commit_and_prover_commitments_reproduce_the_references_vectors(golden):
    """The only fixed vectors upstream that pin a PROVER-side MSM site: vc[j] = PedersenGens::commit(j, r_j), r_j =
    Scalar::random(ChaChaRng::from_seed([24u8; 32])) (tests/range_proof.rs:45-78, 108-113; generators.rs:38-42; the rng is
    restated in oracle/py/chacha_rng.py).  The GPU's commit mirror and the value commitments the GPU prover returns for the
    same openings (RangeProof::prove_multiple, mod.rs:234-288 -> party.rs:52-57) must BE those eight encodings; the proofs it
    makes about them verify against the golden commitments."""
    import chacha_rng
    from bulletproofs_amd import BulletproofGens, RangeProof, Transcript
    r = chacha_rng.golden_blindings()
    vc = [bytes.fromhex(h) for h in golden["value_commitments"]]
    gens = BulletproofGens(64, 8, fixed_window_bits=10)
    pc = gens.pedersen()
    for j in range(8):
        assert gens.commit(j, r[j]) == vc[j], j
    for n, m in ((8, 1), (16, 2), (32, 4), (64, 8)):
        proof, V = RangeProof.prove_multiple(gens, pc, Transcript(golden["label"]), list(range(m)), r[:m], n)
        assert V == vc[:m], (n, m)
        assert proof.verify_multiple(gens, pc, Transcript(golden["label"]), vc[:m], n) is None
    # the constant-time commitment path of the prover (option prover_constant_time) lands on the same encodings
    gens.ctx.set_option("prover_constant_time", 1)
    _, V = RangeProof.prove_multiple(gens, pc, Transcript(golden["label"]), list(range(8)), r, 64)
    assert V == vc

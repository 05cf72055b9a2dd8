"""`transcript: &mut Transcript` at the boundary (src/range_proof/mod.rs:345-353, inner_product_proof.rs:260-270): the
caller's transcript may already hold application messages and is left advanced.  bpgpu_rangeproof_verify_batch_ts takes
the 208-byte STROBE states (one shared by the batch, or one per proof) and returns the advanced ones; every verdict,
mega-check encoding and output state must equal the oracle's verify_ts on the same inputs."""
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


def _bound_state(oracle, i):
    st = oracle.transcript_new(b"payment-protocol v3")
    st = oracle.transcript_append_message(st, b"session", hashlib.shake_256(b"sess%d" % i).digest(40 + i % 7))
    st = oracle.transcript_append_message(st, b"amount-commitment-context", bytes([i & 0xff]) * (i % 5))
    st, _ = oracle.transcript_challenge_bytes(st, b"binding", 16)     # an earlier challenge drawn by the application
    return st


@pytest.mark.parametrize("n,m", [(64, 1), (32, 4), (8, 8)])
def test_per_proof_prebound_transcripts(ctx64x8, oracle, oracle_gens_64_8, n, m):
    nb = 24
    pl = oracle.proof_len(n, m)
    states, proofs, coms = [], b"", b""
    for i in range(nb):
        st = _bound_state(oracle, i)        # every proof has its own transcript history (different `pos` per proof)
        vals = [int.from_bytes(hashlib.shake_256(b"tv%d-%d" % (i, j)).digest(8), "little") % (1 << n) for j in range(m)]
        bl = b"".join(hashlib.shake_256(b"tb%d-%d" % (i, j)).digest(31) + b"\x00" for j in range(m))
        pr, cm, st_after = oracle.prove_ts(oracle_gens_64_8, vals, bl, n, st, b"seed%d" % i)
        states.append(st)
        proofs += pr
        coms += cm
    pb = bytearray(proofs)
    pb[3 * pl + 130] ^= 1                    # wrong t_x
    pb[7 * pl + 165:7 * pl + 192] = b"\xff" * 27   # non-canonical t_x_blinding: FormatError, transcript untouched
    states[11] = _bound_state(oracle, 1000)  # right proof, wrong history
    proofs = bytes(pb)
    rng = hashlib.shake_256(b"ts-rng").digest(64 * nb)
    v, msm, ts_out = ctx64x8.rangeproof_verify_batch_ts(n, m, proofs, pl, coms, b"".join(states), rng, want_msm=True, want_transcripts=True)
    for i in range(nb):
        rc, emsm, est = oracle.verify_ts(oracle_gens_64_8, proofs[pl * i:pl * (i + 1)], coms[32 * m * i:32 * m * (i + 1)], n, states[i],
                                         rng[64 * i:64 * i + 64])
        assert v[i] == rc, i
        if rc in (0, 1) and i != 7:
            assert msm[32 * i:32 * i + 32] == emsm, i
            assert ts_out[208 * i:208 * (i + 1)] == est, i     # prover, oracle verifier and GPU leave the same transcript
        if rc == 2:
            assert ts_out[208 * i:208 * (i + 1)] == states[i]  # from_bytes failed: the transcript was never touched
    assert list(v).count(0) == nb - 3 and v[3] == 1 and v[7] == 2 and v[11] == 1
    # the label-only entry point sees a different statement for every one of them
    v2 = ctx64x8.rangeproof_verify_batch(n, m, proofs, pl, coms, b"payment-protocol v3", rng)
    assert all(x != 0 for x in v2)


def test_shared_prebound_transcript_and_device_pointers(ctx64x8, oracle, oracle_gens_64_8):
    import torch
    import bulletproofs_amd as bp
    L = bp.lib()
    dev = torch.device("cuda", 0)
    n, m, nb = 64, 2, 40
    pl = oracle.proof_len(n, m)
    st = _bound_state(oracle, 5)
    proofs, coms = b"", b""
    for i in range(nb):
        vals = [i + 1, 2 ** 63 + i]
        pr, cm, _ = oracle.prove_ts(oracle_gens_64_8, vals, bytes(range(64)), n, st, b"sh%d" % i)
        proofs += pr
        coms += cm
    pb = bytearray(proofs)
    pb[9 * pl + 3] ^= 8
    proofs = bytes(pb)
    rng = hashlib.shake_256(b"ts-rng2").digest(64 * nb)
    exp = [oracle.verify_ts(oracle_gens_64_8, proofs[pl * i:pl * (i + 1)], coms[64 * i:64 * (i + 1)], n, st, rng[64 * i:64 * i + 64]) for i in range(nb)]
    # host pointers, one shared state (stride 0)
    v, msm, ts_out = ctx64x8.rangeproof_verify_batch_ts(n, m, proofs, pl, coms, st, rng, want_msm=True, want_transcripts=True)
    assert list(v) == [e[0] for e in exp] and v[9] == 1 and list(v).count(0) == nb - 1
    for i in range(nb):
        if i != 9:
            assert msm[32 * i:32 * i + 32] == exp[i][1] and ts_out[208 * i:208 * (i + 1)] == exp[i][2]
    # device pointers, shared state handed over as a host pointer
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_p, d_c, d_r = to_dev(proofs), to_dev(coms), to_dev(rng)
    d_v = torch.full((nb,), 255, dtype=torch.uint8, device=dev)
    d_ts = torch.zeros((nb, 208), dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(device=dev)
    rc = L.bpgpu_rangeproof_verify_batch_ts_dev(ctx64x8.h, n, m, nb, d_p.data_ptr(), pl, d_c.data_ptr(), st, None, d_r.data_ptr(),
                                                d_v.data_ptr(), None, d_ts.data_ptr(), s.cuda_stream)
    assert rc == 0
    s.synchronize()
    assert bytes(d_v.cpu().numpy()) == v and bytes(d_ts.cpu().numpy().tobytes()) == ts_out
    # exactly one transcript source must be given
    assert L.bpgpu_rangeproof_verify_batch_ts_dev(ctx64x8.h, n, m, nb, d_p.data_ptr(), pl, d_c.data_ptr(), None, None, d_r.data_ptr(),
                                                  d_v.data_ptr(), None, None, None) == -1


def test_reference_api_transcript_is_mut(oracle, oracle_gens_64_8):
    """The mirrored RangeProof::verify_multiple advances the caller's Transcript exactly as the oracle's verifier does, so a
    protocol can keep drawing challenges from it afterwards; custom PedersenGens are refused instead of ignored."""
    from bulletproofs_amd import BulletproofGens, PedersenGens, RangeProof, Transcript, VerificationError
    bp_gens = BulletproofGens(64, 8)
    pc_gens = bp_gens.pedersen()
    t_p = oracle.transcript_new(b"app")
    t_p = oracle.transcript_append_message(t_p, b"ctx", b"order 66")
    pr, cm, t_after = oracle.prove_ts(oracle_gens_64_8, [123456789], bytes(32), 64, t_p, b"s")
    t = Transcript(b"app")
    t.append_message(b"ctx", b"order 66")
    assert t.state == t_p
    assert RangeProof.from_bytes(pr).verify_single(bp_gens, pc_gens, t, cm, 64) is None
    assert t.state == t_after
    _, ch_o = oracle.transcript_challenge_bytes(t_after, b"next", 32)
    assert t.challenge_bytes(b"next", 32) == ch_o
    with pytest.raises(VerificationError):   # the advanced transcript is a different statement
        RangeProof.from_bytes(pr).verify_single(bp_gens, pc_gens, t, cm, 64)
    with pytest.raises(VerificationError):
        RangeProof.from_bytes(pr).verify_single(bp_gens, pc_gens, Transcript(b"app"), cm, 64)
    with pytest.raises(ValueError):
        RangeProof.from_bytes(pr).verify_single(bp_gens, PedersenGens(pc_gens.B_blinding, pc_gens.B), Transcript(b"app"), cm, 64)


def test_rejected_before_the_transcript_is_touched(ctx64x8, oracle, oracle_gens_64_8, golden):
    """FormatError (length or scalar), InvalidBitsize, InvalidGeneratorsLength are decided before the reference touches the
    transcript (mod.rs:358-366, 504-538): the caller gets its state back unchanged, for one shared state and for
    per-proof states."""
    case = [x for x in golden["cases"] if x["n"] == 32 and x["m"] == 2][0]
    pr = bytes.fromhex(case["proof"])
    vc = golden["vc_bytes"][:64]
    st = _bound_state(oracle, 3)
    rng = bytes(range(64))
    for states in (st, st + _bound_state(oracle, 4)):          # shared / per proof
        nb = 2
        # malformed length for the whole batch
        v, ts = ctx64x8.rangeproof_verify_batch_ts(32, 2, pr[:-1] * nb, len(pr) - 1, vc * nb, states, rng * nb, want_transcripts=True)
        assert list(v) == [2, 2] and ts == (states if len(states) == 416 else states * 2)
        # bitsize / generator capacity
        v, ts = ctx64x8.rangeproof_verify_batch_ts(24, 2, pr * nb, len(pr), vc * nb, states, rng * nb, want_transcripts=True)
        assert list(v) == [3, 3] and ts == (states if len(states) == 416 else states * 2)
        v, ts = ctx64x8.rangeproof_verify_batch_ts(32, 16, pr * nb, len(pr), golden["vc_bytes"][:32] * 16 * nb, states, rng * nb, want_transcripts=True)
        assert list(v) == [4, 4] and ts == (states if len(states) == 416 else states * 2)
    # and a valid golden proof on a fresh state handed over as state: Ok, state advanced exactly as the oracle's
    fresh = oracle.transcript_new(golden["label"])
    v, ts = ctx64x8.rangeproof_verify_batch_ts(32, 2, pr, len(pr), vc, fresh, rng, want_transcripts=True)
    rc, _, est = oracle.verify_ts(oracle_gens_64_8, pr, vc, 32, fresh, rng)
    assert list(v) == [0] and rc == 0 and ts == est


def test_many_batch_shapes_evict_cached_decompositions(ctx64x8, oracle, oracle_gens_64_8, golden):
    """The per-context cache of work decompositions is bounded (32 entries, least recently used evicted): 40 different batch
    sizes, then the first ones again -- every verdict still equals the oracle's."""
    case = [x for x in golden["cases"] if x["n"] == 64 and x["m"] == 1][0]
    pr = bytes.fromhex(case["proof"])
    bad = bytearray(pr)
    bad[130] ^= 4
    vc = golden["vc_bytes"][:32]
    for nb in list(range(1, 41)) + [1, 2, 3]:
        proofs = b"".join(bytes(bad) if i % 3 == 1 else pr for i in range(nb))
        v = ctx64x8.rangeproof_verify_batch(64, 1, proofs, len(pr), vc * nb, golden["label"], bytes(64 * nb))
        assert list(v) == [1 if i % 3 == 1 else 0 for i in range(nb)], nb

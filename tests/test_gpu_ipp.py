"""GPU parity for the stand-alone inner-product proof entry point bpgpu_ipp_verify_batch
(InnerProductProof::from_bytes + verify, src/inner_product_proof.rs:260-326, 373-407); the cases mirror the
reference's own tests (ipp.rs:433-534: test_helper_create(n) for n in {1, 2, 4, 32, 64})."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import bulletproofs_amd as bp
    c = bp.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n", [1, 2, 4, 32, 64])
def test_make_ipp_verify(ctx, oracle, n):
    nb = 6
    insts = [oracle.ipp_test_instance(n, b"innerproducttest", b"gipp-%d-%d" % (n, j)) for j in range(nb)]
    pl = len(insts[0]["proof"])
    t = bytearray(insts[1]["proof"])
    t[-40] ^= 1                                                   # tampered a
    insts[1] = dict(insts[1], proof=bytes(t))
    insts[2] = dict(insts[2], P=insts[2]["Q"])                    # wrong P
    nc = bytearray(insts[3]["proof"])
    nc[-32:] = b"\xff" * 32                                       # b not canonical -> FormatError (ipp.rs:401-404)
    insts[3] = dict(insts[3], proof=bytes(nc))
    if n > 1:
        li = bytearray(insts[4]["proof"])
        li[0:32] = bytes(32)                                      # L_0 = identity -> VerificationError (transcript.rs:75-87)
        insts[4] = dict(insts[4], proof=bytes(li))
    cat = lambda key: b"".join(i[key] for i in insts)
    verdict, msm = ctx.ipp_verify_batch(n, cat("proof"), pl, b"innerproducttest", cat("Gf"), cat("Hf"), cat("P"), cat("Q"), cat("G"), cat("H"),
                                        want_msm=True)
    for j, inst in enumerate(insts):
        rc, em = oracle.ipp_verify(n, inst["proof"], b"innerproducttest", inst["Gf"], inst["Hf"], inst["P"], inst["Q"], inst["G"], inst["H"])
        assert verdict[j] == rc, (n, j)
        if rc != 2 and em != b"\xff" * 32 and not (n > 1 and j == 4):
            assert msm[32 * j:32 * j + 32] == em, (n, j)
    assert list(verdict) == [0, 1, 1, 2, (1 if n > 1 else 0), 0]
    # wrong label, wrong n, malformed length
    v = ctx.ipp_verify_batch(n, insts[0]["proof"], pl, b"other", insts[0]["Gf"], insts[0]["Hf"], insts[0]["P"], insts[0]["Q"], insts[0]["G"], insts[0]["H"])
    assert list(v) == [1 if n > 1 else 0]    # n = 1 has no challenge: the label does not enter the check
    v = ctx.ipp_verify_batch(2 * n, insts[0]["proof"], pl, b"innerproducttest", insts[0]["Gf"] * 2, insts[0]["Hf"] * 2, insts[0]["P"], insts[0]["Q"],
                             insts[0]["G"] * 2, insts[0]["H"] * 2)
    assert list(v) == [1]
    v = ctx.ipp_verify_batch(n, insts[0]["proof"][:-1], pl - 1, b"x", insts[0]["Gf"], insts[0]["Hf"], insts[0]["P"], insts[0]["Q"], insts[0]["G"], insts[0]["H"])
    assert list(v) == [2]


@pytest.mark.parametrize("n", [2, 64])
def test_ipp_device_pointers_shared_bases_and_prebound_transcript(ctx, oracle, n):
    """bpgpu_ipp_verify_batch_dev: device pointers, G / H given ONCE for the batch (the reference's callers pass the same
    generator vectors for every proof, ipp.rs:260-270), and a caller transcript that already holds messages."""
    import torch
    import bulletproofs_amd as bp
    L = bp.lib()
    dev = torch.device("cuda", 0)
    inst = oracle.ipp_test_instance(n, b"innerproducttest", b"shared-%d" % n)
    pl = len(inst["proof"])
    bad = bytearray(inst["proof"])
    bad[-40] ^= 1
    proofs = inst["proof"] + bytes(bad) + inst["proof"] + inst["proof"]
    P = inst["P"] * 2 + inst["Q"] + inst["P"]
    nb = 4
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d = {k: to_dev(v) for k, v in dict(pr=proofs, Gf=inst["Gf"] * nb, Hf=inst["Hf"] * nb, P=P, Q=inst["Q"] * nb, G=inst["G"], H=inst["H"]).items()}
    d_v = torch.full((nb,), 255, dtype=torch.uint8, device=dev)
    d_o = torch.zeros((nb, 32), dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(device=dev)
    rc = L.bpgpu_ipp_verify_batch_dev(ctx.h, n, nb, d["pr"].data_ptr(), pl, b"innerproducttest", 16, None, d["Gf"].data_ptr(), d["Hf"].data_ptr(),
                                      d["P"].data_ptr(), d["Q"].data_ptr(), d["G"].data_ptr(), d["H"].data_ptr(), 1, d_v.data_ptr(), d_o.data_ptr(),
                                      s.cuda_stream)
    assert rc == 0
    s.synchronize()
    got, out = bytes(d_v.cpu().numpy()), d_o.cpu().numpy().tobytes()
    for j in range(nb):
        erc, em = oracle.ipp_verify(n, proofs[pl * j:pl * (j + 1)], b"innerproducttest", inst["Gf"], inst["Hf"], P[32 * j:32 * j + 32], inst["Q"],
                                    inst["G"], inst["H"])
        assert got[j] == erc and out[32 * j:32 * j + 32] == em, (n, j)
    assert list(got) == [0, 1, 1, 0]
    # a transcript with history: the fresh-label proof no longer verifies against it, and it equals the label path when
    # the state is exactly Transcript::new(label)
    from bulletproofs_amd._lib import transcript_new, transcript_append_message
    st = transcript_new(b"innerproducttest")
    for state, expect in ((st, [0, 1, 1, 0]), (transcript_append_message(st, b"x", b"y"), [1, 1, 1, 1])):
        rc = L.bpgpu_ipp_verify_batch_dev(ctx.h, n, nb, d["pr"].data_ptr(), pl, None, 0, state, d["Gf"].data_ptr(), d["Hf"].data_ptr(),
                                          d["P"].data_ptr(), d["Q"].data_ptr(), d["G"].data_ptr(), d["H"].data_ptr(), 1, d_v.data_ptr(), None,
                                          s.cuda_stream)
        assert rc == 0
        s.synchronize()
        assert list(d_v.cpu().numpy()) == expect


def test_verification_scalars_export_vs_oracle(oracle):
    """bpgpu_ipp_verification_scalars == InnerProductProof::verification_scalars (src/inner_product_proof.rs:198-253; the call of
    src/r1cs/verifier.rs:401-404): u_i^2, u_i^-2, s_i, status and the advanced transcript against the oracle, for the sizes of the
    reference's tests, the R1CS shape's n = 2048, label / shared-state / per-proof-state transcripts, and the three error kinds."""
    import bulletproofs_amd as bp
    from bulletproofs_amd._lib import transcript_new, transcript_append_message
    ctx = bp.Context(0)
    label = b"innerproducttest"
    for n in (1, 2, 4, 32, 64, 2048):
        nb = 5 if n < 2048 else 2
        insts = [oracle.ipp_test_instance(n, label, b"gv%d-%d" % (n, i))["proof"] for i in range(nb)]
        pl = len(insts[0])
        bad = [bytearray(p_) for p_ in insts]
        if n >= 2:
            bad[1][0:32] = bytes(32)                      # identity L_0 -> VerificationError
        bad[nb - 1][pl - 32:pl] = b"\xff" * 32            # b not canonical -> FormatError
        proofs = b"".join(bytes(x) for x in bad)
        st0 = transcript_new(label)
        us, ui, s_, st, tso = ctx.ipp_verification_scalars(n, proofs, pl, label=label, want_transcripts=True)
        k = n.bit_length() - 1
        for i in range(nb):
            rc, eus, eui, es, est = oracle.ipp_verification_scalars(n, bytes(bad[i]), st0)
            assert st[i] == rc, (n, i)
            assert tso[208 * i:208 * (i + 1)] == est, (n, i, rc)          # on every path: an identity L_0 leaves the domain separator in, a malformed proof nothing
            if rc == 0:
                assert us[32 * k * i:32 * k * (i + 1)] == eus and ui[32 * k * i:32 * k * (i + 1)] == eui and s_[32 * n * i:32 * n * (i + 1)] == es
            else:
                assert s_[32 * n * i:32 * n * (i + 1)] == bytes(32 * n)
        # the caller's transcript: one state for the batch, then one per proof (different positions)
        shared = transcript_append_message(st0, b"earlier", b"message of the parent protocol")
        us2, ui2, s2, st2, tso2 = ctx.ipp_verification_scalars(n, proofs, pl, transcripts=shared, want_transcripts=True)
        per = b"".join(transcript_append_message(st0, b"p", bytes(j for j in range(3 * i + 1))) for i in range(nb))
        us3, ui3, s3, st3, tso3 = ctx.ipp_verification_scalars(n, proofs, pl, transcripts=per, want_transcripts=True)
        for i in range(nb):
            for (gu, gi, gs, gst, gts, start) in ((us2, ui2, s2, st2, tso2, shared), (us3, ui3, s3, st3, tso3, per[208 * i:208 * (i + 1)])):
                rc, eus, eui, es, est = oracle.ipp_verification_scalars(n, bytes(bad[i]), start)
                assert gst[i] == rc
                assert gts[208 * i:208 * (i + 1)] == est, (n, i, rc)
                if rc == 0:
                    assert gu[32 * k * i:32 * k * (i + 1)] == eus and gi[32 * k * i:32 * k * (i + 1)] == eui and gs[32 * n * i:32 * n * (i + 1)] == es
        # n that does not match the proof: VerificationError for well-formed proofs, FormatError outranks it
        _, _, _, st4, tso4 = ctx.ipp_verification_scalars(2 * n, proofs, pl, label=label, want_transcripts=True)
        assert list(st4) == [2 if i == nb - 1 else 1 for i in range(nb)]
        assert tso4 == st0 * nb                                            # ipp.rs:203-211: Err before the domain separator -- the transcript as it came
        # an identity at every position of the proof: the state as of that message (ipp.rs:217-222)
        if 2 <= n <= 64:
            stops = [insts[0][:32 * u] + bytes(32) + insts[0][32 * u + 32:] for u in range(2 * k)]
            _, _, _, st6, tso6 = ctx.ipp_verification_scalars(n, b"".join(stops), pl, transcripts=per[:208], want_transcripts=True)
            for u in range(2 * k):
                rc, _, _, _, est = oracle.ipp_verification_scalars(n, stops[u], per[:208])
                assert st6[u] == rc == 1 and tso6[208 * u:208 * (u + 1)] == est, (n, u)
    _, _, _, st5 = ctx.ipp_verification_scalars(4, bytes(100), 50, label=label)      # malformed length
    assert list(st5) == [2, 2]
    ctx.close()

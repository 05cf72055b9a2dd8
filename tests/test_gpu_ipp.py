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

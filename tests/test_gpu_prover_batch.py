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

"""the PROVER's multiscalar multiplications through the same engine.

InnerProductProof::create (src/inner_product_proof.rs:38-193) computes, per round, L and R as MSMs of size
2n'+1 (ipp.rs:87-131) and folds the generator vectors with n' MSMs of size 2 (ipp.rs:153-178).  The test
runs the pure-Python twin of that routine twice -- once with its own big-integer MSM, once with every MSM sent
to bpgpu_msm_batch (compressed points in, compressed point out) -- and requires byte-identical proofs; the
result must also verify through bpgpu_ipp_verify_batch.  (The reference uses constant-time MSMs at the sites
that touch secrets; the engine is variable-time, so this is a parity/throughput demonstration, as 8f-4 says.)"""
import hashlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "py"))


@pytest.mark.parametrize("n", [4, 16])
def test_ipp_create_with_gpu_msms_is_byte_identical(n):
    import bp_twin as T
    import bulletproofs_amd as bp

    ctx = bp.Context(0)
    L = T.L

    def rnd(tag):
        return int.from_bytes(hashlib.shake_256(tag).digest(64), "little") % L

    gens = T.BulletproofGens(n, 1)
    G, H = gens.G(n, 1), gens.H(n, 1)
    Q = T.from_uniform_bytes(hashlib.shake_256(b"Q").digest(64))
    a = [rnd(b"a%d" % i) for i in range(n)]
    b = [rnd(b"b%d" % i) for i in range(n)]
    y_inv = T.sc_inv(rnd(b"y"))
    Gf = [1] * n
    Hf = [pow(y_inv, i, L) for i in range(n)]

    calls = []

    def gpu_msm(scalars, points):
        sb = b"".join((s % L).to_bytes(32, "little") for s in scalars)
        pb = b"".join(T.compress(p) for p in points)
        out, st = ctx.msm_batch([len(scalars)], sb, pb)
        assert st == b"\x00"
        calls.append(len(scalars))
        return T.decompress(out)

    ref = T.ipp_create(T.Transcript(b"innerproducttest"), Q, Gf, Hf, G, H, a, b)
    cpu_msm = T.msm
    T.msm = gpu_msm
    try:
        got = T.ipp_create(T.Transcript(b"innerproducttest"), Q, Gf, Hf, G, H, a, b)
    finally:
        T.msm = cpu_msm
    assert got == ref
    lg = n.bit_length() - 1
    assert sorted(set(calls), reverse=True)[0] == n + 1 and len(calls) == sum(2 + 2 * (n >> (r + 1)) for r in range(lg))

    # and the proof verifies on the GPU: P = <a,G'> + <b',H'> + <a,b> Q with the factors folded in (ipp.rs:433-497)
    L_vec, R_vec, fa, fb = got
    proof = b"".join(l + r for l, r in zip(L_vec, R_vec)) + fa.to_bytes(32, "little") + fb.to_bytes(32, "little")
    bprime = [x * h % L for x, h in zip(b, Hf)]
    c = sum(x * y for x, y in zip(a, b)) % L
    P = T.msm(a + bprime + [c], list(G) + list(H) + [Q])
    enc = lambda v: b"".join(x.to_bytes(32, "little") for x in v)
    verdict = ctx.ipp_verify_batch(n, proof, len(proof), b"innerproducttest", enc(Gf), enc(Hf), T.compress(P), T.compress(Q),
                                   b"".join(T.compress(p) for p in G), b"".join(T.compress(p) for p in H))
    assert verdict == b"\x00"
    ctx.close()

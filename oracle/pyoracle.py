"""ctypes loader for liboracle.so (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

ERR_NAMES = {0: "Ok", 1: "VerificationError", 2: "FormatError", 3: "InvalidBitsize",
             4: "InvalidGeneratorsLength", 5: "InvalidAggregation"}


def build(force=False):
    srcs = [os.path.join(_HERE, "c", f) for f in os.listdir(os.path.join(_HERE, "c"))] + \
           [os.path.join(_HERE, "bp_oracle.h"), os.path.join(_HERE, "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def build_native():
    """bench.py's cpu_baseline leg: rebuild the oracle with -march=native on the box it is timed on (the committed
    Makefile targets x86-64-v3 so that the checker built in the authoring container runs anywhere).  Must be called
    before the library is first loaded; returns True when the native build is the one in use."""
    global _SO
    if _lib is not None:
        return _SO.endswith("liboracle_native.so")
    so = os.path.join(_HERE, "liboracle_native.so")
    try:
        srcs = [os.path.join(_HERE, "c", f) for f in ("ge.c", "merlin.c", "bp.c")]
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=gnu11", "-w", "-shared", "-o", so] + srcs + ["-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _SO = so
        return True
    except Exception:
        return False


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    u8p, sz, vp = C.c_char_p, C.c_size_t, C.c_void_p
    L.oracle_gens_new.restype = vp
    L.oracle_gens_new.argtypes = [sz, sz]
    L.oracle_gens_free.argtypes = [vp]
    L.oracle_gens_export.argtypes = [vp, u8p, u8p, u8p, u8p]
    L.oracle_point_decompress_ok.argtypes = [u8p]
    L.oracle_from_uniform_bytes.argtypes = [u8p, u8p]
    L.oracle_scalar_from_wide.argtypes = [u8p, u8p]
    L.oracle_scalar_mul.argtypes = [u8p, u8p, u8p]
    L.oracle_scalar_invert.argtypes = [u8p, u8p]
    L.oracle_merlin_kat.argtypes = [u8p, sz, u8p, u8p, sz, u8p, u8p, sz]
    L.oracle_shake256.argtypes = [u8p, sz, u8p, sz]
    L.oracle_sha3_512.argtypes = [u8p, sz, u8p]
    L.oracle_msm.argtypes = [sz, u8p, u8p, C.c_int, u8p]
    L.oracle_last_msm_ops.restype = C.c_uint64
    L.oracle_verify.argtypes = [vp, u8p, sz, u8p, sz, sz, u8p, sz, u8p, u8p]
    L.oracle_verify_ts.argtypes = [vp, u8p, sz, u8p, sz, sz, u8p, u8p, u8p]
    L.oracle_transcript_new.argtypes = [u8p, sz, u8p]
    L.oracle_transcript_append_message.argtypes = [u8p, u8p, u8p, sz]
    L.oracle_transcript_challenge_bytes.argtypes = [u8p, u8p, u8p, sz]
    L.oracle_verify_terms.argtypes = [vp, u8p, sz, u8p, sz, sz, u8p, sz, u8p, u8p, u8p, C.POINTER(sz)]
    L.oracle_prove.argtypes = [vp, C.POINTER(C.c_uint64), u8p, sz, sz, u8p, sz, u8p, sz, u8p, u8p]
    L.oracle_prove_ts.argtypes = [vp, C.POINTER(C.c_uint64), u8p, sz, sz, u8p, u8p, sz, u8p, u8p]
    L.oracle_verify_batch.restype = C.c_double
    L.oracle_verify_batch.argtypes = [vp, sz, u8p, sz, u8p, sz, sz, u8p, sz, u8p, u8p, u8p, C.c_int]
    L.oracle_prove_batch.restype = C.c_double
    L.oracle_prove_batch.argtypes = [vp, sz, C.POINTER(C.c_uint64), u8p, sz, sz, u8p, sz, u8p, sz, u8p, u8p, C.c_int]
    L.oracle_ipp_verify.argtypes = [sz, u8p, sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
    L.oracle_ipp_verification_scalars.argtypes = [sz, u8p, sz, u8p, u8p, u8p, u8p]
    L.oracle_backend.restype = C.c_char_p
    L.oracle_ipp_test_instance.argtypes = [sz, u8p, sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
    L.oracle_ipp_create.argtypes = [sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
    L.oracle_prove_shares.argtypes = [vp, C.POINTER(C.c_uint64), u8p, sz, sz, u8p, sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p]
    L.oracle_audit_share.argtypes = [vp, sz, sz, u8p, u8p, u8p, u8p, u8p]
    L.oracle_linear_create.argtypes = [sz, u8p, u8p, u8p, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
    L.oracle_linear_verify.argtypes = [sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
    L.oracle_msm_batch.restype = C.c_double
    L.oracle_msm_batch.argtypes = [sz, sz, u8p, u8p, C.c_int, u8p, u8p, C.c_int]
    _lib = L
    return L


def backend():
    """which field backend this build's Straus MSM / point decoding use ("u64 5x51 serial", or the SIMD one of a -march=native build)"""
    return lib().oracle_backend().decode()


def proof_len(n, m):
    return 32 * (9 + 2 * ((n * m).bit_length() - 1))


def n_terms(n, m):
    return 2 * n * m + 2 * ((n * m).bit_length() - 1) + m + 6


class Gens:
    """BulletproofGens::new(gens_capacity, party_capacity) + PedersenGens::default()."""

    def __init__(self, gens_capacity, party_capacity):
        self.gens_capacity, self.party_capacity = gens_capacity, party_capacity
        self.h = lib().oracle_gens_new(gens_capacity, party_capacity)

    def __del__(self):
        try:
            lib().oracle_gens_free(self.h)
        except Exception:
            pass

    def export(self):
        tot = self.gens_capacity * self.party_capacity
        G = C.create_string_buffer(32 * tot)
        H = C.create_string_buffer(32 * tot)
        B = C.create_string_buffer(32)
        Bb = C.create_string_buffer(32)
        lib().oracle_gens_export(self.h, G, H, B, Bb)
        return G.raw, H.raw, B.raw, Bb.raw


def msm(scalars, points, algo=0):
    """scalars/points: bytes of n*32 each. Returns (status, 32-byte result)."""
    n = len(scalars) // 32
    assert len(points) == 32 * n
    out = C.create_string_buffer(32)
    st = lib().oracle_msm(n, scalars, points, algo, out)
    return st, out.raw


def verify(gens, proof, commitments, n, label, rng64):
    m = len(commitments) // 32
    out = C.create_string_buffer(32)
    rc = lib().oracle_verify(gens.h, proof, len(proof), commitments, m, n, label, len(label), rng64, out)
    return rc, out.raw


def transcript_new(label):
    st = C.create_string_buffer(208)
    lib().oracle_transcript_new(label, len(label), st)
    return st.raw


def transcript_append_message(state, label, msg):
    st = C.create_string_buffer(state, 208)
    lib().oracle_transcript_append_message(st, label, msg, len(msg))
    return st.raw


def transcript_challenge_bytes(state, label, n):
    st = C.create_string_buffer(state, 208)
    out = C.create_string_buffer(n)
    lib().oracle_transcript_challenge_bytes(st, label, out, n)
    return st.raw, out.raw


def verify_ts(gens, proof, commitments, n, state, rng64):
    """verify_multiple_with_rng with the caller's transcript state (208 bytes); returns (rc, msm encoding, advanced state)."""
    m = len(commitments) // 32
    out = C.create_string_buffer(32)
    st = C.create_string_buffer(state, 208)
    rc = lib().oracle_verify_ts(gens.h, proof, len(proof), commitments, m, n, st, rng64, out)
    return rc, out.raw, st.raw


def verify_terms(gens, proof, commitments, n, label, rng64):
    m = len(commitments) // 32
    N = n_terms(n, m) if n * m > 0 else 0
    sc = C.create_string_buffer(32 * max(N, 1))
    pt = C.create_string_buffer(32 * max(N, 1))
    nt = C.c_size_t(0)
    rc = lib().oracle_verify_terms(gens.h, proof, len(proof), commitments, m, n, label, len(label), rng64,
                                   sc, pt, C.byref(nt))
    if rc:
        return rc, b"", b""
    return 0, sc.raw[:32 * nt.value], pt.raw[:32 * nt.value]


def prove(gens, values, blindings, n, label, seed):
    m = len(values)
    vals = (C.c_uint64 * m)(*values)
    proof = C.create_string_buffer(proof_len(n, m))
    com = C.create_string_buffer(32 * m)
    rc = lib().oracle_prove(gens.h, vals, blindings, m, n, label, len(label), seed, len(seed), proof, com)
    if rc:
        raise ValueError("oracle_prove failed: %s" % ERR_NAMES.get(rc, rc))
    return proof.raw, com.raw


def prove_ts(gens, values, blindings, n, state, seed):
    """prove_multiple_with_rng on a caller-supplied transcript state; returns (proof, commitments, advanced state)."""
    m = len(values)
    vals = (C.c_uint64 * m)(*values)
    proof = C.create_string_buffer(proof_len(n, m))
    com = C.create_string_buffer(32 * m)
    st = C.create_string_buffer(state, 208)
    rc = lib().oracle_prove_ts(gens.h, vals, blindings, m, n, st, seed, len(seed), proof, com)
    if rc:
        raise ValueError("oracle_prove_ts failed: %s" % ERR_NAMES.get(rc, rc))
    return proof.raw, com.raw, st.raw


def prove_batch(gens, values, blindings, m, n, label, seed, threads=1):
    nb = len(values) // m
    vals = (C.c_uint64 * len(values))(*values)
    pl = proof_len(n, m)
    proofs = C.create_string_buffer(pl * nb)
    com = C.create_string_buffer(32 * m * nb)
    lib().oracle_prove_batch(gens.h, nb, vals, blindings, m, n, label, len(label), seed, len(seed), proofs, com, threads)
    return proofs.raw, com.raw


def verify_batch(gens, proofs, commitments, m, n, label, rng64s, threads=1):
    pl = proof_len(n, m)
    nb = len(proofs) // pl
    verdicts = C.create_string_buffer(nb)
    outs = C.create_string_buffer(32 * nb)
    secs = lib().oracle_verify_batch(gens.h, nb, proofs, pl, commitments, m, n, label, len(label), rng64s,
                                     verdicts, outs, threads)
    return secs, verdicts.raw, outs.raw


def msm_batch(nbatch, n, scalars, points, algo=0, threads=1):
    outs = C.create_string_buffer(32 * nbatch)
    status = C.create_string_buffer(nbatch)
    secs = lib().oracle_msm_batch(nbatch, n, scalars, points, algo, outs, status, threads)
    return secs, outs.raw, status.raw


def ipp_test_instance(n, label, seed):
    """The reference's test_helper_create(n) (inner_product_proof.rs:433-497): returns a dict of byte strings."""
    lg = n.bit_length() - 1
    bufs = dict(proof=C.create_string_buffer(32 * (2 * lg + 2)), P=C.create_string_buffer(32), Q=C.create_string_buffer(32),
                G=C.create_string_buffer(32 * n), H=C.create_string_buffer(32 * n), Gf=C.create_string_buffer(32 * n),
                Hf=C.create_string_buffer(32 * n))
    rc = lib().oracle_ipp_test_instance(n, label, len(label), seed, len(seed), bufs["proof"], bufs["P"], bufs["Q"], bufs["G"],
                                        bufs["H"], bufs["Gf"], bufs["Hf"])
    assert rc == 0
    return {k: v.raw for k, v in bufs.items()}


def ipp_create(n, label, Q, Hf, G, H, a, b):
    """InnerProductProof::create(...).to_bytes() with G_factors = 1 (ipp.rs:38-193)."""
    lg = n.bit_length() - 1
    out = C.create_string_buffer(32 * (2 * lg + 2))
    rc = lib().oracle_ipp_create(n, label, len(label), Q, Hf, G, H, a, b, out)
    return rc, out.raw


def ipp_verification_scalars(n, proof, state):
    """InnerProductProof::verification_scalars on a 208-byte transcript state: (rc, u_sq, u_inv_sq, s, advanced state)."""
    k = (len(proof) // 32 - 2) // 2 if len(proof) >= 64 else 0
    us, ui, s_ = C.create_string_buffer(32 * max(k, 1)), C.create_string_buffer(32 * max(k, 1)), C.create_string_buffer(32 * max(n, 1))
    st = C.create_string_buffer(state, 208)
    rc = lib().oracle_ipp_verification_scalars(n, proof, len(proof), st, us, ui, s_)
    return rc, us.raw[:32 * k], ui.raw[:32 * k], s_.raw[:32 * n], st.raw


def ipp_verify(n, proof, label, Gf, Hf, P, Q, G, H):
    out = C.create_string_buffer(32)
    rc = lib().oracle_ipp_verify(n, proof, len(proof), label, len(label), Gf, Hf, P, Q, G, H, out)
    return rc, out.raw


def linear_create(n, state, rng, Cc, r, a, b, G, F, B):
    """LinearProof::create(transcript, rng, &C, r, a, b, G, &F, &B).to_bytes() (linear_proof.rs:40-173); rng = 64 bytes per
    Scalar::random in draw order.  Returns (rc, proof bytes)."""
    lg = max(n, 1).bit_length() - 1
    assert len(rng) >= 64 * (2 * lg + 2)
    out = C.create_string_buffer(32 * (2 * lg + 3))
    rc = lib().oracle_linear_create(n, state, rng, Cc, r, a, b, G, F, B, out)
    return rc, out.raw


def linear_verify(n, proof, state, Cc, G, F, B, b):
    """LinearProof::from_bytes(proof)?.verify(transcript, &C, &G, &F, &B, b) (linear_proof.rs:175-236): (code, compress(expect_S - S))"""
    out = C.create_string_buffer(32)
    rc = lib().oracle_linear_verify(n, proof, len(proof), state, Cc, G, F, B, b, out)
    return rc, out.raw


def linear_test_instance(n, seed, label=b"linearprooftest"):
    """The reference's test_helper(n) (linear_proof.rs:401-466) with a SHAKE256(seed) rng: G = bp_gens.share(0).G(n),
    F = pedersen B, B = pedersen B_blinding, random a, b, r, C = <a, G> + r B + <a, b> F.  Returns a dict of byte strings."""
    import hashlib
    lg = n.bit_length() - 1
    g = Gens(n, 1)
    Gc, _, Bp, Bb = g.export()
    stream = hashlib.shake_256(seed).digest(64 * (2 * n + 1 + 2 * lg + 2))
    L = lib()

    def wide(i):
        o = C.create_string_buffer(32)
        L.oracle_scalar_from_wide(stream[64 * i:64 * i + 64], o)
        return o.raw
    a = b"".join(wide(i) for i in range(n))
    b = b"".join(wide(n + i) for i in range(n))
    r = wide(2 * n)
    ell = 2 ** 252 + 27742317777372353535851937790883648493
    c = sum(int.from_bytes(a[32 * i:32 * i + 32], "little") * int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(n)) % ell
    rcm, Cc = msm(a + r + c.to_bytes(32, "little"), Gc[:32 * n] + Bb + Bp)
    assert rcm == 0
    st = transcript_new(label)
    rng = stream[64 * (2 * n + 1):]
    rc, proof = linear_create(n, st, rng, Cc, r, a, b, Gc[:32 * n], Bp, Bb)
    assert rc == 0
    return dict(n=n, proof=proof, C=Cc, G=Gc[:32 * n], F=Bp, B=Bb, a=a, b=b, r=r, rng=rng, label=label)


def prove_shares(gens, values, blindings, n, label, seed):
    """prove_multiple as parties + dealer run it, with the messages they exchange (messages.rs): returns a dict with proof,
    commitments, bit_commitments (m x 96), poly_commitments (m x 64), shares (m x 32 (3 + 2n)), challenges (y, z, x)."""
    m = len(values)
    lg = (n * m).bit_length() - 1
    bufs = dict(proof=C.create_string_buffer(32 * (9 + 2 * lg)), commitments=C.create_string_buffer(32 * m), bit_commitments=C.create_string_buffer(96 * m),
                poly_commitments=C.create_string_buffer(64 * m), shares=C.create_string_buffer(32 * (3 + 2 * n) * m), challenges=C.create_string_buffer(96))
    va = (C.c_uint64 * m)(*values)
    rc = lib().oracle_prove_shares(gens.h, va, blindings, m, n, label, len(label), seed, len(seed), bufs["proof"], bufs["commitments"],
                                   bufs["bit_commitments"], bufs["poly_commitments"], bufs["shares"], bufs["challenges"])
    assert rc == 0, rc
    return {k: v.raw for k, v in bufs.items()}


def audit_share(gens, n, j, share, bit_commitment, poly_commitment, challenges):
    """ProofShare::audit_share (messages.rs:85-167): (0 Ok / 1 Err, compress(P_check) + compress(t_check))"""
    out = C.create_string_buffer(64)
    rc = lib().oracle_audit_share(gens.h, n, j, share, bit_commitment, poly_commitment, challenges, out)
    return rc, out.raw

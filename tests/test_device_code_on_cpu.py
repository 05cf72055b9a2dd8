"""The per-lane bodies of the HIP kernels (bulletproofs_amd/csrc/*.h) compiled for the host with
bounds checks (tests/cpu_harness) and compared with the oracle / Python twin.  Covers the exact
device arithmetic and pipeline logic without a GPU; the `-m gpu` tests then only have to show that
the same code behaves identically when launched as kernels."""
import ctypes as C
import hashlib
import os
import random
import sys

import pytest

import bp_twin as T
import harness_lib


@pytest.fixture(scope="module")
def H():
    return harness_lib.lib()


def _b(x):
    return (x % T.P).to_bytes(32, "little")


def _feop(H, op, x, y=0):
    out = C.create_string_buffer(32)
    H.h_fe_op(op, _b(x), _b(y), out)
    return int.from_bytes(out.raw, "little")


def test_field_arithmetic_matches_big_ints(H):
    P = T.P
    random.seed(7)
    vals = [0, 1, 2, P - 1, P - 2, 19, 2**255 - 20, 2**26 - 1, 2**51] + [random.randrange(P) for _ in range(200)]
    for i, x in enumerate(vals):
        y = vals[(i * 7 + 3) % len(vals)]
        assert _feop(H, 0, x, y) == x * y % P
        assert _feop(H, 1, x) == x * x % P
        assert _feop(H, 2, x, y) == (x + y) % P
        assert _feop(H, 3, x, y) == (x - y) % P
        assert _feop(H, 4, x) == (-x) % P
        assert _feop(H, 8, x, y) == ((2 * x + y) * (x + 2 * y) - (2 * x + y) ** 2) % P   # lazy-limb chain
        if i < 24:
            assert _feop(H, 5, x) == pow(x, P - 2, P)
            assert _feop(H, 6, x) == pow(x, (P - 5) // 8, P)
        fl = H.h_fe_flags(_b(x))
        assert (fl & 1) == (x % P & 1) and bool(fl & 2) == (x % P == 0)
    out = C.create_string_buffer(32)   # non-canonical input bytes are reduced implicitly
    H.h_fe_op(7, (P + 5).to_bytes(32, "little"), _b(0), out)
    assert int.from_bytes(out.raw, "little") == 5


def test_group_ops_and_ristretto_codec(H):
    pts = [T.from_uniform_bytes(hashlib.shake_256(b"p%d" % i).digest(64)) for i in range(24)]
    enc = [T.compress(p) for p in pts]
    o = C.create_string_buffer(32)
    for i in range(24):
        a, b = enc[i], enc[(i + 1) % 24]
        pa, pb = pts[i], pts[(i + 1) % 24]
        xy = C.create_string_buffer(128)
        assert H.h_decompress(a, xy) == 1
        d = T.decompress(a)
        assert [int.from_bytes(xy.raw[32 * k:32 * k + 32], "little") for k in range(4)] == list(d)
        H.h_compress_xyzt(xy, o)
        assert o.raw == a
        exp = {0: T.pt_add(pa, pb), 1: T.pt_add(pa, T.pt_neg(pb)), 2: T.pt_dbl(pa), 3: T.pt_add(pa, pb),
               4: T.pt_add(pa, T.pt_neg(pb)), 5: pa, 6: T.pt_add(T.pt_mul(16, pa), pb), 7: T.pt_neg(pa)}
        for op, e in exp.items():
            assert H.h_point_op(op, a, b, o) & 1 and o.raw == T.compress(e), (op, i)
        assert H.h_point_op(1, a, a, o) == 3 and o.raw == bytes(32)      # P - P: identity flag + zero encoding
        u = hashlib.shake_256(b"u%d" % i).digest(64)
        H.h_from_uniform(u, o)
        assert o.raw == T.compress(T.from_uniform_bytes(u))
    P = T.P
    for e in [bytes([1]) + bytes(31), P.to_bytes(32, "little"), (P + 2).to_bytes(32, "little"), b"\xff" * 32, (2).to_bytes(32, "little")]:
        assert (H.h_decompress(e, C.create_string_buffer(128)) == 1) == (T.decompress(e) is not None)
    for i in range(200):   # random byte strings: validity decision must agree with RFC 9496 decoding
        e = hashlib.shake_256(b"e%d" % i).digest(32)
        e = bytes([e[0] & 0xfe]) + e[1:31] + bytes([e[31] & 0x7f])
        assert (H.h_decompress(e, C.create_string_buffer(128)) == 1) == (T.decompress(e) is not None)
    assert H.h_decompress(bytes(32), C.create_string_buffer(128)) == 1


def test_scalar_field_keccak_merlin(H):
    L = T.L
    random.seed(3)

    def scop(op, x, y=0):
        o = C.create_string_buffer(32)
        H.h_sc_op(op, x.to_bytes(32, "little"), y.to_bytes(32, "little"), o)
        return int.from_bytes(o.raw, "little")
    vals = [0, 1, 2, L - 1, L - 2, 2**252, 2**128] + [random.randrange(L) for _ in range(60)]
    for i, x in enumerate(vals):
        y = vals[(3 * i + 1) % len(vals)]
        assert scop(0, x, y) == x * y % L and scop(1, x, y) == (x + y) % L
        assert scop(2, x, y) == (x - y) % L and scop(3, x) == (-x) % L
        if i < 8 and x:
            assert scop(4, x) == pow(x, L - 2, L)
        # division-step inversion (scinv.h): same canonical result as the l-2 exponentiation, 0 -> 0
        assert scop(7, x) == pow(x, L - 2, L) and scop(8, x) == pow(x, L - 2, L) and scop(9, x) == pow(x, L - 2, L)
        lo, hi = random.randrange(2**256), random.randrange(2**256)
        assert scop(5, lo, hi) == (lo + (hi << 256)) % L
    assert scop(5, 2**256 - 1, 2**256 - 1) == (2**512 - 1) % L
    for x in [3, 2**30, 2**30 - 1, 2**60 + 1, (L - 1) // 2, (L + 1) // 2, 2**252 - 1, 2**252 + 1] + [random.randrange(L) for _ in range(400)] \
            + [random.randrange(2**b) for b in range(1, 253, 3)] + [L - random.randrange(1, 2**b) for b in range(1, 250, 5)]:
        assert scop(7, x) == pow(x, L - 2, L), hex(x)
        assert scop(9, x) == pow(x, L - 2, L), hex(x)   # the variable-time form narrow chains run (round 6)
    for x in [random.randrange(L) for _ in range(3000)]:
        assert scop(9, x) == pow(x, L - 2, L), hex(x)
    out = C.create_string_buffer(32)
    H.h_merlin_kat(b"test protocol", 13, b"some label", 10, b"some data", 9, b"challenge", 9, out, 32)
    assert out.raw.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    for n in (0, 1, 71, 72, 135, 136, 137, 300):
        msg = (bytes(range(256)) * 2)[:n]
        o = C.create_string_buffer(300)
        H.h_shake256(msg, n, o, 300)
        assert o.raw == hashlib.shake_256(msg).digest(300)
        o = C.create_string_buffer(64)
        H.h_sha3_512(msg, n, o)
        assert o.raw == hashlib.sha3_512(msg).digest()


def _sc(tag):
    return (int.from_bytes(hashlib.shake_256(tag).digest(64), "little") % T.L).to_bytes(32, "little")


def _pt(oracle, tag):
    o = C.create_string_buffer(32)
    oracle.lib().oracle_from_uniform_bytes(hashlib.shake_256(tag).digest(64), o)
    return o.raw


def test_variable_base_pipeline_lane_by_lane(H, oracle):
    sizes = [0, 1, 2, 31, 32, 33, 70]
    S = P = b""
    for k, n in enumerate(sizes):
        S += b"".join(_sc(b"m%d-s%d" % (k, i)) for i in range(n))
        P += b"".join(_pt(oracle, b"m%d-p%d" % (k, i)) for i in range(n))
    nt = (C.c_uint32 * len(sizes))(*sizes)
    out, st = C.create_string_buffer(32 * len(sizes)), C.create_string_buffer(len(sizes))
    H.h_msm_vb(len(sizes), nt, S, P, out, st)
    off = 0
    for k, n in enumerate(sizes):
        assert st.raw[k] == 0 and out.raw[32 * k:32 * k + 32] == oracle.msm(S[off:off + 32 * n], P[off:off + 32 * n])[1]
        off += 32 * n
    sp = [0, 1, T.L - 1, 8, int("8" * 63, 16) % T.L, 2**252]
    s = b"".join(x.to_bytes(32, "little") for x in sp)
    p = b"".join(_pt(oracle, b"sp%d" % i) for i in range(len(sp)))
    nt = (C.c_uint32 * 1)(len(sp))
    H.h_msm_vb(1, nt, s, p, out, st)
    assert out.raw[:32] == oracle.msm(s, p)[1] and st.raw[0] == 0
    bad = bytearray(p)
    bad[0] |= 1
    H.h_msm_vb(1, nt, s, bytes(bad), out, st)
    assert st.raw[0] == 1 and out.raw[:32] == bytes(32)
    s2 = bytearray(s)
    s2[0:32] = T.L.to_bytes(32, "little")
    H.h_msm_vb(1, nt, bytes(s2), p, out, st)
    assert st.raw[0] == 2


def test_variable_base_pipeline_narrow_form_lane_by_lane(H, oracle):
    """The narrow form of the small-MSM pipeline (option msm_narrow: k_vb_prepare_hi / k_vb_window_hi / k_vb_tail_narrow): chunks of ~sqrt(N)
    terms, second tables of the points' 2^128 multiples (wavefront-cooperative decode + 128 cooperative doublings + the cooperative table),
    32-window chain, encoding through the split inverse square root -- lane for lane on the host, against the oracle."""
    sizes = [0, 1, 2, 31, 33, 70]
    S = P = b""
    for k, n in enumerate(sizes):
        S += b"".join(_sc(b"n%d-s%d" % (k, i)) for i in range(n))
        P += b"".join(_pt(oracle, b"n%d-p%d" % (k, i)) for i in range(n))
    nt = (C.c_uint32 * len(sizes))(*sizes)
    out, st = C.create_string_buffer(32 * len(sizes)), C.create_string_buffer(len(sizes))
    for chunk, levels in ((4, 2), (9, 2), (32, 2), (9, 4)):   # (levels 4: tables of 2^64 P, 2^128 P, 2^192 P, a 16-window chain)
        H.h_msm_vb_narrow(len(sizes), nt, chunk, levels, S, P, out, st)
        off = 0
        for k, n in enumerate(sizes):
            assert st.raw[k] == 0 and out.raw[32 * k:32 * k + 32] == oracle.msm(S[off:off + 32 * n], P[off:off + 32 * n])[1], (chunk, levels, k)
            off += 32 * n
    sp = [0, 1, T.L - 1, 8, int("8" * 63, 16) % T.L, 2**252, 2**128, 2**128 - 1, (2**124 - 1) << 128]
    s = b"".join(x.to_bytes(32, "little") for x in sp)
    p = b"".join(_pt(oracle, b"nsp%d" % i) for i in range(len(sp)))
    nt = (C.c_uint32 * 1)(len(sp))
    for levels in (2, 4):
        H.h_msm_vb_narrow(1, nt, 4, levels, s, p, out, st)
        assert out.raw[:32] == oracle.msm(s, p)[1] and st.raw[0] == 0
    bad = bytearray(p)
    bad[0] |= 1
    H.h_msm_vb_narrow(1, nt, 4, 4, s, bytes(bad), out, st)
    assert st.raw[0] == 1 and out.raw[:32] == bytes(32)


@pytest.mark.parametrize("n,m", [(8, 1), (8, 2), (16, 4)])
def test_batched_rangeproof_prover_lane_by_lane(H, oracle, n, m):
    """rp_prover.h: RangeProof::prove_multiple_with_rng (mod.rs:234-288; party.rs, dealer.rs) with every commitment as a
    multiscalar multiplication over the generator tables and the inner-product argument of ipp_prover.h -- proofs,
    commitments and final transcripts byte-identical to the oracle's prover given the same random scalars (the oracle draws
    them from SHAKE256(seed): the same bytes are handed to the device code), and the oracle's verifier accepts."""
    nb = 3
    g = oracle.Gens(n, m)
    G, Hh, B, Bb = g.export()
    gens = Bb + B + G + Hh
    vals = [int.from_bytes(hashlib.shake_256(b"pv%d-%d-%d" % (n, m, i)).digest(8), "little") % (1 << n) for i in range(nb * m)]
    bl = b"".join(hashlib.shake_256(b"pb%d-%d-%d" % (n, m, i)).digest(32) for i in range(nb * m))   # arbitrary 32 bytes: reduced mod l
    per = 64 * (m * (2 * n + 2) + 2 * m)
    seeds = [b"rpp-%d-%d-%d" % (n, m, p) for p in range(nb)]
    rng = b"".join(hashlib.shake_256(sd).digest(per) for sd in seeds)
    st0 = oracle.transcript_append_message(oracle.transcript_new(b"prover"), b"ctx", b"hello")
    pl = oracle.proof_len(n, m)
    proofs, coms, ts = C.create_string_buffer(pl * nb), C.create_string_buffer(32 * m * nb), C.create_string_buffer(208 * nb)
    va = (C.c_uint64 * len(vals))(*vals)
    assert H.h_rp_prove(4, n, m, gens, n, m, nb, va, bl, st0, rng, proofs, coms, ts) == 0
    for p in range(nb):
        epr, ecm, ets = oracle.prove_ts(g, vals[p * m:(p + 1) * m], bl[32 * m * p:32 * m * (p + 1)], n, st0, seeds[p])
        assert coms.raw[32 * m * p:32 * m * (p + 1)] == ecm, (n, m, p)
        assert proofs.raw[pl * p:pl * (p + 1)] == epr, (n, m, p)
        assert ts.raw[208 * p:208 * (p + 1)] == ets
        assert oracle.verify_ts(g, epr, ecm, n, st0, bytes(64))[0] == 0


@pytest.mark.parametrize("n", [1, 2, 8, 32])
def test_batched_ipp_prover_lane_by_lane(H, oracle, n):
    """ipp_prover.h (InnerProductProof::create without folding the generators: every L_j / R_j is one MSM over the original
    points) produces byte-identical proofs to the oracle's restatement of ipp.rs:38-193, accepted by the oracle's verifier;
    per-proof and batch-shared bases; general G_factors against the Python twin."""
    nb = 3
    inst = [oracle.ipp_test_instance(n, b"innerproducttest", b"prv-%d-%d" % (n, j)) for j in range(nb)]
    a = [b"".join(_sc(b"pa%d-%d-%d" % (n, j, i)) for i in range(n)) for j in range(nb)]
    b = [b"".join(_sc(b"pb%d-%d-%d" % (n, j, i)) for i in range(n)) for j in range(nb)]
    cat = lambda k: b"".join(x[k] for x in inst)
    pl = 32 * (2 * (n.bit_length() - 1) + 2)
    ts0 = oracle.transcript_new(b"innerproducttest")
    out, st = C.create_string_buffer(pl * nb), C.create_string_buffer(nb)
    assert H.h_ipp_create(n, nb, ts0, cat("Q"), cat("Gf"), cat("Hf"), cat("G"), cat("H"), 0, b"".join(a), b"".join(b), out, st) == 0
    for j in range(nb):
        rc, exp = oracle.ipp_create(n, b"innerproducttest", inst[j]["Q"], inst[j]["Hf"], inst[j]["G"], inst[j]["H"], a[j], b[j])
        assert rc == 0 and st.raw[j] == 0 and out.raw[pl * j:pl * (j + 1)] == exp, (n, j)
    # shared bases: every proof over instance 0's generators
    out2 = C.create_string_buffer(pl * nb)
    assert H.h_ipp_create(n, nb, ts0, cat("Q"), cat("Gf"), cat("Hf"), inst[0]["G"], inst[0]["H"], 1, b"".join(a), b"".join(b), out2, st) == 0
    for j in range(nb):
        rc, exp = oracle.ipp_create(n, b"innerproducttest", inst[j]["Q"], inst[j]["Hf"], inst[0]["G"], inst[0]["H"], a[j], b[j])
        assert out2.raw[pl * j:pl * (j + 1)] == exp, (n, j)
    if 1 < n <= 8:   # general G_factors, and a transcript with history: the Python twin
        gf = b"".join(_sc(b"gf%d-%d" % (n, i)) for i in range(n))
        tw = T.Transcript(b"app")
        tw.append_message(b"ctx", b"prover test")
        ts1 = oracle.transcript_append_message(oracle.transcript_new(b"app"), b"ctx", b"prover test")
        sc = lambda bs: [int.from_bytes(bs[32 * i:32 * i + 32], "little") for i in range(len(bs) // 32)]
        pts = lambda bs: [T.decompress(bs[32 * i:32 * i + 32]) for i in range(len(bs) // 32)]
        exp = T.ipp_create(tw, T.decompress(inst[0]["Q"]), sc(gf), sc(inst[0]["Hf"]), pts(inst[0]["G"]), pts(inst[0]["H"]), sc(a[0]), sc(b[0]))
        out3, st3 = C.create_string_buffer(pl), C.create_string_buffer(1)
        assert H.h_ipp_create(n, 1, ts1, inst[0]["Q"], gf, inst[0]["Hf"], inst[0]["G"], inst[0]["H"], 0, a[0], b[0], out3, st3) == 0
        Lv, Rv, af, bf = exp
        exp_bytes = b"".join(l + r for l, r in zip(Lv, Rv)) + af.to_bytes(32, "little") + bf.to_bytes(32, "little")
        assert out3.raw == exp_bytes


@pytest.mark.parametrize("c", [8, 12])
def test_bucket_msm_pipeline_phase_by_phase(H, oracle, c):
    """bucket.h (Pippenger: counting sort by digit, population-sorted buckets, running-sum tree, Horner over the window
    sums) with the kernels' own per-lane phase functions: ragged sizes incl. empty MSMs, edge scalars, repeated and
    negated points (complete addition formulas), undecodable point / non-canonical scalar, and the single-MSM mode with
    skipped "proofs" of the batch combination."""
    sizes = [0, 1, 2, 70, 300] if c == 8 else [0, 3, 150]
    S = P = b""
    for k, n in enumerate(sizes):
        S += b"".join(_sc(b"b%d-s%d" % (k, i)) for i in range(n))
        P += b"".join(_pt(oracle, b"b%d-p%d" % (k, i % 40)) for i in range(n))   # points repeat: buckets see P + P and P - P
    nt = (C.c_uint32 * len(sizes))(*sizes)
    out, st = C.create_string_buffer(32 * len(sizes)), C.create_string_buffer(len(sizes))
    assert H.h_msm_bucket(len(sizes), nt, S, P, c, None, 0, 1, out, st) == 0
    out_b = C.create_string_buffer(32 * len(sizes))
    assert H.h_msm_bucket(len(sizes), nt, S, P, c, None, 0, 3, out_b, st) == 0 and out_b.raw == out.raw   # split sort (large MSMs)
    off = 0
    for k, n in enumerate(sizes):
        assert st.raw[k] == 0 and out.raw[32 * k:32 * k + 32] == oracle.msm(S[off:off + 32 * n], P[off:off + 32 * n])[1], (c, k)
        off += 32 * n
    sp = [0, 1, T.L - 1, 8, int("8" * 63, 16) % T.L, 2**252, 2**252 + 1, (1 << (c - 1)), (1 << (c - 1)) - 1, (1 << c) - 1, T.L - (1 << (c - 1))]
    s = b"".join(x.to_bytes(32, "little") for x in sp)
    p = b"".join(_pt(oracle, b"bsp%d" % (i % 3)) for i in range(len(sp)))
    nt1 = (C.c_uint32 * 1)(len(sp))
    assert H.h_msm_bucket(1, nt1, s, p, c, None, 0, 1, out, st) == 0
    assert out.raw[:32] == oracle.msm(s, p)[1] and st.raw[0] == 0
    bad = bytearray(p)
    bad[0] |= 1
    H.h_msm_bucket(1, nt1, s, bytes(bad), c, None, 0, 1, out, st)
    assert st.raw[0] == 1 and out.raw[:32] == bytes(32)
    s2 = bytearray(s)
    s2[0:32] = T.L.to_bytes(32, "little")
    H.h_msm_bucket(1, nt1, bytes(s2), p, c, None, 0, 1, out, st)
    assert st.raw[0] == 2
    # single-MSM mode: 6 "proofs" of 5 terms each, proofs 1 and 4 rejected -> their terms stay out of the combination
    n = 30
    s = b"".join(_sc(b"rl-s%d" % i) for i in range(n))
    p = b"".join(_pt(oracle, b"rl-p%d" % i) for i in range(n))
    nts = (C.c_uint32 * 6)(*[5] * 6)
    skip = (C.c_uint32 * 6)(0, 1, 0, 0, 2, 0)
    assert H.h_msm_bucket(6, nts, s, p, c, skip, 5, 2, out, st) == 0
    keep = [i for i in range(n) if i // 5 not in (1, 4)]
    assert out.raw[:32] == oracle.msm(b"".join(s[32 * i:32 * i + 32] for i in keep), b"".join(p[32 * i:32 * i + 32] for i in keep))[1]


@pytest.mark.parametrize("lanes", [1, 7, 64, 128, 256])
def test_fused_bucket_chain_phase_by_phase(H, oracle, lanes):
    """bucket2.h (round 6: digits as window-major bytes, sort into an LDS list, equal runs of list entries per lane whatever the bucket
    populations are, head pieces combined by the lane that owns the bucket) with the kernels' own per-lane phase functions.  Every
    stage is checked on the way (the short-register decode == the plain one, the list is the window's non-zero digits sorted by
    magnitude) and the encodings against the oracle: ragged sizes incl. empty MSMs and MSMs with fewer terms than lanes, edge scalars,
    repeated and negated points, an undecodable point, a non-canonical scalar.  lanes = 1 and 7 are not device widths: they make one
    lane's run cross many buckets / buckets of a small input span several lanes."""
    sizes = [0, 1, 2, 70, 300, 517]
    S = P = b""
    for k, n in enumerate(sizes):
        S += b"".join(_sc(b"f%d-s%d" % (k, i)) for i in range(n))
        P += b"".join(_pt(oracle, b"f%d-p%d" % (k, i % 40)) for i in range(n))
    nt = (C.c_uint32 * len(sizes))(*sizes)
    out, st, stats = C.create_string_buffer(32 * len(sizes)), C.create_string_buffer(len(sizes)), (C.c_uint32 * 4)()
    assert H.h_msm_bucket2(len(sizes), nt, S, P, lanes, out, st, stats) == 0
    off = 0
    for k, n in enumerate(sizes):
        assert st.raw[k] == 0 and out.raw[32 * k:32 * k + 32] == oracle.msm(S[off:off + 32 * n], P[off:off + 32 * n])[1], (lanes, k)
        off += 32 * n
    if lanes == 1:
        assert stats[0] == 0 and stats[1] == 0                    # one lane: every bucket begins and ends inside its run
    if lanes == 7:
        assert stats[0] > 0 and stats[1] > 0                      # buckets do span lanes
    sp = [0, 1, T.L - 1, 8, int("8" * 63, 16) % T.L, 2**252, 2**252 + 1, 128, 127, 255, T.L - 128, 129, 256, 2**248]
    s = b"".join(x.to_bytes(32, "little") for x in sp)
    p = b"".join(_pt(oracle, b"fsp%d" % (i % 3)) for i in range(len(sp)))
    nt1 = (C.c_uint32 * 1)(len(sp))
    assert H.h_msm_bucket2(1, nt1, s, p, lanes, out, st, None) == 0
    assert out.raw[:32] == oracle.msm(s, p)[1] and st.raw[0] == 0
    bad = bytearray(p)
    bad[0] |= 1
    H.h_msm_bucket2(1, nt1, s, bytes(bad), lanes, out, st, None)
    assert st.raw[0] == 1 and out.raw[:32] == bytes(32)
    s2 = bytearray(s)
    s2[0:32] = T.L.to_bytes(32, "little")
    H.h_msm_bucket2(1, nt1, bytes(s2), p, lanes, out, st, None)
    assert st.raw[0] == 2


@pytest.mark.parametrize("lanes", [8, 64])
def test_fused_bucket_chain_crowded_buckets_spread_over_lanes(H, oracle, lanes):
    """What bucket.h needed a heavy pass for: scalars that put many terms into one bucket of a window (all equal, all small, 128-bit
    ones sharing their high windows, five values).  In the fused chain a crowded bucket is spread over the lanes like any other run of
    list entries; the lane in which it begins adds the other lanes' head pieces.  Results == oracle, and the statistics say that the
    long chains of head pieces were really walked (equal scalars: one bucket per middle window holds every term -- every other lane's
    run is a head piece behind one owner)."""
    def run(scal, pts):
        n = len(scal) // 32
        out, st, stats = C.create_string_buffer(32), C.create_string_buffer(1), (C.c_uint32 * 4)()
        assert H.h_msm_bucket2(1, (C.c_uint32 * 1)(n), scal, pts, lanes, out, st, stats) == 0
        assert st.raw[0] == 0 and out.raw == oracle.msm(scal, pts)[1]
        return list(stats)
    n = 400
    pts = b"".join(_pt(oracle, b"hv-p%d" % (i % 57)) for i in range(n))
    eq = _sc(b"hv-equal") * n
    q = -(-n // lanes)
    assert run(eq, pts)[2] == -(-n // q) - 1                  # every lane with entries but the first hands a head piece to the owner
    small = b"".join((1 + (i % 3)).to_bytes(32, "little") for i in range(n))
    assert run(small, pts)[2] >= 2
    short = b"".join(int.from_bytes(_sc(b"hv-w%d" % i)[:16], "little").to_bytes(32, "little") for i in range(n))
    run(short, pts)
    many = b"".join(_sc(b"hv-m%d" % (i % 5)) for i in range(n))
    assert run(many, pts)[2] >= 2
    for m in (1, 2, lanes - 1, lanes, lanes + 1, 2 * lanes + 1):             # runs of zero, one, two entries per lane
        run(_sc(b"hv-one") * m, pts[:32 * m])
        run(b"".join(_sc(b"hv-r%d" % i) for i in range(m)), pts[:32 * m])


@pytest.mark.parametrize("c", [8, 12])
def test_bucket_msm_crowded_buckets_go_through_the_heavy_pass(H, oracle, c):
    """bucket.h stage 3b: a lane adds its whole bucket up to `lim` terms; of a more crowded one it adds lim - 32 and the rest goes
    64 lanes at a time (k_bk_heavy: G wavefronts per window list the crowded buckets among their ranks, strided partial sums, tree
    through LDS, add to the bucket's sum).  Scalars that defeat bk_recode's stirring -- all equal, all small, short ones sharing
    their high windows -- with the lowest limit (33) so that a few hundred terms reach every branch (one lane busy / all lanes /
    several rounds; several crowded buckets per wavefront), then the natural limit.  Results == oracle; the pass reports how many
    buckets it took."""
    def run(scal, pts, lim):
        n = len(scal) // 32
        out, st = C.create_string_buffer(32), C.create_string_buffer(1)
        H.h_set_bucket_cap(lim)
        try:
            assert H.h_msm_bucket(1, (C.c_uint32 * 1)(n), scal, pts, c, None, 0, 1, out, st) == 0
        finally:
            H.h_set_bucket_cap(0)
        assert st.raw[0] == 0 and out.raw == oracle.msm(scal, pts)[1]
        return H.h_bucket_heavy_count()
    n = 400
    pts = b"".join(_pt(oracle, b"hv-p%d" % (i % 57)) for i in range(n))
    eq = _sc(b"hv-equal") * n
    assert run(eq, pts, 33) > 0
    assert run(eq, pts, 0) > 0        # natural limit (2 x 400 / half + 32): the windows in bits 135 .. 251 hold all 400 terms in ONE bucket each
    small = b"".join((1 + (i % 3)).to_bytes(32, "little") for i in range(n))
    assert run(small, pts, 34) > 0
    short = b"".join(int.from_bytes(_sc(b"hv-w%d" % i)[:16], "little").to_bytes(32, "little") for i in range(n))   # 128-bit scalars: the round's finding
    run(short, pts, 33)
    for extra in (1, 31, 32, 33, 63, 64, 65, 129):                           # one bucket of lim + extra terms: rest = 32 + extra
        m = 40 + extra
        assert run(_sc(b"hv-one") * m, pts[:32 * m], 40) > 0
    rnd = b"".join(_sc(b"hv-r%d" % i) for i in range(n))
    assert run(rnd, pts, 0) == 0
    many = b"".join(_sc(b"hv-m%d" % (i % 5)) for i in range(n))             # five crowded buckets of 80 per middle window
    assert run(many, pts, 33) >= 5
    for G in (1, 2, 16):                                                     # any partition of the ranks over wavefronts
        H.h_set_bucket_groups(G)
        try:
            assert run(many, pts, 33) >= 5 and run(eq, pts, 0) > 0
        finally:
            H.h_set_bucket_groups(0)


@pytest.mark.parametrize("W,nsplit", [(4, 3), (5, 8), (7, 1)])
def test_shared_generator_pipeline_lane_by_lane(H, oracle, W, nsplit):
    g = oracle.Gens(8, 2)
    G, Hh, B, Bb = g.export()
    gens = Bb + B + G + Hh
    n, m = 8, 2
    ngen = 2 * n * m + 2
    ids = [0, 1] + [2 + j * 8 + i for j in range(m) for i in range(n)] + [2 + 16 + j * 8 + i for j in range(m) for i in range(n)]
    nb, nu = 4, 35
    GS = b"".join(_sc(b"g%d-%d-%d" % (W, b, i)) for b in range(nb) for i in range(ngen))
    edge = [0, 1, T.L - 1, 1 << (W - 1), (1 << W) - 1, 2**252]
    GS = b"".join(x.to_bytes(32, "little") for x in edge) + GS[32 * len(edge):]
    US = b"".join(_sc(b"us%d-%d" % (b, i)) for b in range(nb) for i in range(nu))
    UP = b"".join(_pt(oracle, b"up%d-%d" % (b, i)) for b in range(nb) for i in range(nu))
    out, st, vd = C.create_string_buffer(32 * nb), C.create_string_buffer(nb), C.create_string_buffer(nb)
    assert H.h_msm_shared(W, nsplit, 34, gens, ngen, (C.c_uint32 * ngen)(*ids), nb, nu, GS, US, UP, out, st, vd) == 0
    gp = b"".join(gens[32 * i:32 * i + 32] for i in ids)
    for b in range(nb):
        exp = oracle.msm(GS[32 * ngen * b:32 * ngen * (b + 1)] + US[32 * nu * b:32 * nu * (b + 1)], gp + UP[32 * nu * b:32 * nu * (b + 1)])
        assert st.raw[b] == 0 and out.raw[32 * b:32 * b + 32] == exp[1] and vd.raw[b] == 1
    assert H.h_msm_shared(W, nsplit, 34, gens, ngen, (C.c_uint32 * ngen)(*ids), 1, 0, bytes(32 * ngen), b"", b"", out, st, vd) == 0
    assert vd.raw[0] == 0 and out.raw[:32] == bytes(32)


@pytest.mark.parametrize("horner_lanes", [1, 4, 64, "radix32", "radix32+a_outside", "a_outside", "64+defer_emit", "64+split", "64+defer_emit+split", "64+defer_emit+split+hi", "64+defer_emit+split+hi4"])
def test_full_verification_pipeline_lane_by_lane_on_golden_proofs(H, oracle, golden, horner_lanes):
    """rp_transcript -> rp_expand_a/b -> vb_* / fb_* -> finish, emulated lane by lane, on the reference's
    golden proofs (small shapes; the GPU tests cover all 16) plus tampered copies.  The Horner layouts:
    1 lane per chain (msm_vb.h), 4 (horner_quad.h), 64 (horner_wave.h); the wide chains' forms (one-lane chain): "radix32" -- the proofs'
    own points in signed radix 32 (16-entry tables, 51 windows); "a_outside" -- A, whose coefficient is 1, added after the chain."""
    defer = isinstance(horner_lanes, str) and "defer_emit" in horner_lanes   # the narrow chain's form: the scalar role's coefficients parked by the leader, recoded by 32 lanes (rp_defer)
    split = isinstance(horner_lanes, str) and "split" in horner_lanes        # ... and its split scalar role (round 6): k + 1 lanes invert one value each, the basepoint coefficients are a role of the next launch
    H.h_set_defer_emit(1 if defer else 0)
    H.h_set_coop_split(1 if split else 0)
    H.h_set_narrow_hi((4 if "+hi4" in horner_lanes else 2) if isinstance(horner_lanes, str) and "+hi" in horner_lanes else 0)   # ... and for very narrow ones second tables of the points' 2^128 multiples, a 32-window chain
    if defer or split:
        horner_lanes = 64
    wide = isinstance(horner_lanes, str)
    H.h_set_radix5(1 if wide and "radix32" in horner_lanes else 0)
    H.h_set_a_outside(1 if wide and "a_outside" in horner_lanes else 0)
    H.h_set_horner_lanes(1 if wide else horner_lanes)
    label = golden["label"]
    vc = golden["vc_bytes"]
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        if n * m > 32:
            continue
        pr = bytes.fromhex(case["proof"])
        bad = bytearray(pr)
        bad[128] ^= 1
        proofs = pr + bytes(bad) + pr
        coms = vc[:32 * m] + vc[:32 * m] + ((vc[32:32 * m] + vc[:32]) if m > 1 else vc[32:64])
        rng = hashlib.shake_256(b"h%d%d" % (n, m)).digest(64 * 3)
        gg = oracle.Gens(n, m)
        G2, H2, B2, Bb2 = gg.export()
        vd, mo = C.create_string_buffer(3), C.create_string_buffer(96)
        assert H.h_rp_verify(4, 3, n, m, Bb2 + B2 + G2 + H2, n, m, 3, proofs, len(pr), coms, label, len(label), rng, vd, mo) == 0
        for b in range(3):
            erc, emsm = oracle.verify(gg, proofs[len(pr) * b:len(pr) * (b + 1)], coms[32 * m * b:32 * m * (b + 1)], n, label, rng[64 * b:64 * b + 64])
            assert vd.raw[b] == erc and mo.raw[32 * b:32 * b + 32] == emsm, (n, m, b)
        assert list(vd.raw) == [0, 1, 1]
    pr = bytes.fromhex(golden["cases"][0]["proof"])
    nc = bytearray(pr)
    nc[128:160] = b"\xff" * 32
    ia = bytearray(pr)
    ia[0:32] = bytes(32)
    us = bytearray(pr)
    us[32] |= 1
    gg = oracle.Gens(8, 1)
    G2, H2, B2, Bb2 = gg.export()
    vd, mo = C.create_string_buffer(4), C.create_string_buffer(128)
    rng = hashlib.shake_256(b"zz").digest(256)
    assert H.h_rp_verify(4, 2, 8, 1, Bb2 + B2 + G2 + H2, 8, 1, 4, bytes(nc) + bytes(ia) + bytes(us) + pr, len(pr), vc[:32] * 4, label,
                         len(label), rng, vd, mo) == 0
    assert list(vd.raw) == [2, 1, 1, 0]
    H.h_set_radix5(0)
    H.h_set_a_outside(0)
    H.h_set_defer_emit(0)
    H.h_set_coop_split(0)
    H.h_set_narrow_hi(0)


def test_radix32_recoding_reconstructs_the_scalar(H):
    """sc_recode32 / sc_digit32 (msm_vb.h): 51 digits in [-16, 15] with sum_w d_w 32^w == s, for edge scalars and random ones."""
    import random
    L = 2**252 + 27742317777372353535851937790883648493
    rnd = random.Random(5)
    cases = [0, 1, 15, 16, 17, 31, 32, L - 1, L - 2, 2**252, 2**252 - 1, (2**255 - 1) // 31 % L] + [rnd.randrange(L) for _ in range(200)]
    for s in cases:
        dg = (C.c_int32 * 51)()
        H.h_recode32(s.to_bytes(32, "little"), dg)
        assert all(-16 <= d <= 15 for d in dg)
        assert sum(d * 32**w for w, d in enumerate(dg)) == s


def test_coalesced_launch_segment_table_lane_by_lane(H, oracle, golden):
    """bpgpu_pool_*'s coalesced launches (rp_seg, csrc/rangeproof.h): the proofs of one launch come from several submitted
    items, each with its own proof / commitment / rng buffers (every second item without rng bytes of its own) and its own
    verdict / encoding buffers.  Lane by lane: same results as the contiguous launch and as the oracle."""
    H.h_set_horner_lanes(4)
    label = golden["label"]
    vc = golden["vc_bytes"]
    case = [c for c in golden["cases"] if c["n"] == 8 and c["m"] == 2][0]
    n, m = 8, 2
    pr = bytes.fromhex(case["proof"])
    nb = 19
    plist = []
    for b in range(nb):
        q = bytearray(pr)
        if b in (2, 5):
            q[128] ^= 1 << b                                   # VerificationError
        if b == 4:
            q[160:192] = b"\xff" * 32                           # FormatError
        plist.append(bytes(q))
    proofs = b"".join(plist)
    coms = b"".join(vc[:32 * m] if b != 6 else vc[32:32 * m] + vc[:32] for b in range(nb))
    rng = hashlib.shake_256(b"segs").digest(64 * nb)
    gg = oracle.Gens(n, m)
    G2, H2, B2, Bb2 = gg.export()
    exp = [oracle.verify(gg, plist[b], coms[32 * m * b:32 * m * (b + 1)], n, label, rng[64 * b:64 * b + 64]) for b in range(nb)]
    assert [e[0] for e in exp] == [0, 0, 1, 0, 2, 1, 1] + [0] * 12
    for sizes in ([], [1] * 19, [1] * 15 + [4], [3, 16], [2, 1, 3, 1, 12], [18, 1], [19]):   # > 16 items: table in memory, else inline
        H.h_set_segments(len(sizes), (C.c_uint32 * max(len(sizes), 1))(*sizes))
        vd, mo = C.create_string_buffer(nb), C.create_string_buffer(32 * nb)
        assert H.h_rp_verify(4, 3, n, m, Bb2 + B2 + G2 + H2, n, m, nb, proofs, len(pr), coms, label, len(label), rng, vd, mo) == 0
        for b in range(nb):
            assert vd.raw[b] == exp[b][0] and mo.raw[32 * b:32 * b + 32] == exp[b][1], (sizes, b)
    # items that differ only in their LABEL (one length: every transcript position is shared) ride in one chain, each from its own start
    # state (rp_seg::init_w): the mega-check encodings under a foreign label are the oracle's, the items under the right label still verify
    labels = [label, bytes(reversed(label)), bytes((c + 1) & 0xff for c in label)]
    H.h_set_segment_labels(3, b"".join(labels), len(label))
    for sizes in ([2, 1, 3, 1, 12], [1] * 19):
        H.h_set_segments(len(sizes), (C.c_uint32 * len(sizes))(*sizes))
        vd, mo = C.create_string_buffer(nb), C.create_string_buffer(32 * nb)
        assert H.h_rp_verify(4, 3, n, m, Bb2 + B2 + G2 + H2, n, m, nb, proofs, len(pr), coms, label, len(label), rng, vd, mo) == 0
        b, n_ok = 0, 0
        for j, cnt in enumerate(sizes):
            for _ in range(cnt):
                e = oracle.verify(gg, plist[b], coms[32 * m * b:32 * m * (b + 1)], n, labels[j % 3], rng[64 * b:64 * b + 64])
                assert vd.raw[b] == e[0] and mo.raw[32 * b:32 * b + 32] == e[1], (sizes, b)
                assert (e[0] == 0) == (j % 3 == 0 and exp[b][0] == 0)
                n_ok += e[0] == 0
                b += 1
        assert b == nb and n_ok > 0
    H.h_set_segment_labels(0, b"", 0)
    H.h_set_segments(0, (C.c_uint32 * 1)(0))


def test_scripted_transcript_equals_bytewise_replay_and_oracle(H, oracle, golden):
    """rp_script.h: the verifier transcript compiled into per-shape XOR masks + record positions (launch 1's transcript role)
    against the byte-wise STROBE replay, lane by lane -- per-proof scalars, status words, advanced transcripts -- for all golden
    shapes, tampered copies, and start states at every rate position class (records straddling the 166-byte rate boundary,
    a permutation falling exactly at the end of a record, the domain separator applied on the device or not); the byte-wise
    path's final transcript is also the oracle's."""
    from bulletproofs_amd._lib import transcript_new, transcript_append_message
    label = golden["label"]
    vc = golden["vc_bytes"]
    seen_perms = set()
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        pr = bytes.fromhex(case["proof"])
        k = (n * m).bit_length() - 1
        t0 = bytearray(pr)
        t0[128] ^= 1                        # tampered scalar: transcript still runs
        z = bytearray(pr)
        z[224:256] = bytes(32)              # L_0 = identity encoding -> VerificationError from the transcript
        f = bytearray(pr)
        f[160:192] = b"\xff" * 32          # FormatError: transcript untouched
        proofs = pr + bytes(t0) + bytes(z) + bytes(f)
        nb = 4
        coms = vc[:32 * m] * nb
        rng = hashlib.shake_256(b"ts%d%d" % (n, m)).digest(64 * nb)
        for pad in ([], [0], [1], [7], [60], [100, 3], [165], [166], [167, 9], [40, 40, 40]):
            for domsep in (1, 0):
                st = transcript_new(label)
                for j, ln in enumerate(pad):
                    st = transcript_append_message(st, b"pad%d" % j, bytes((ln + q) & 0xff for q in range(ln)))
                if not domsep:
                    st = transcript_append_message(st, b"dom-sep", b"rangeproof v1")
                    st = transcript_append_message(st, b"n", n.to_bytes(8, "little"))
                    st = transcript_append_message(st, b"m", m.to_bytes(8, "little"))
                nperm = C.c_uint32(0)
                so, to = C.create_string_buffer(nb), C.create_string_buffer(208 * nb)
                rc = H.h_rp_transcript_compare(n, m, nb, proofs, len(pr), coms, rng, st, domsep, C.byref(nperm), so, to)
                assert rc == 0, (n, m, pad, domsep, rc)
                assert list(so.raw) == [0, 0, 1, 2]
                seen_perms.add(nperm.value)
                if len(pad) < 2:    # the oracle's verifier on the same start state ends in the same transcript (valid proof and tampered scalar)
                    st_o = st if domsep else None
                    if domsep:
                        for b_ in (0, 1):
                            rc_o, _, ts_o = oracle.verify_ts(oracle.Gens(n, m), proofs[len(pr) * b_:len(pr) * (b_ + 1)], coms[:32 * m], n, st_o, rng[64 * b_:64 * b_ + 64])
                            assert ts_o == to.raw[208 * b_:208 * (b_ + 1)], (n, m, pad, b_)
    assert len(seen_perms) > 3


def test_scripted_transcript_with_one_start_state_per_proof(H, oracle, golden):
    """The combining queue of the pool hands a chain one caller transcript PER PROOF, all at the same STROBE position
    (bpgpu_pool_rangeproof_verify_ts): the scripted replay then starts from ts_in[p] instead of a common state.  Lane by lane
    against the byte-wise replay from the same states, and against the oracle's verify_ts (advanced transcript) per proof."""
    from bulletproofs_amd._lib import transcript_new, transcript_append_message
    vc = golden["vc_bytes"]
    for case in golden["cases"][::3]:
        n, m = case["n"], case["m"]
        pr = bytes.fromhex(case["proof"])
        nb = 5
        f = bytearray(pr)
        f[160:192] = b"\xff" * 32          # FormatError: this proof's transcript is handed back untouched
        proofs = pr * 3 + bytes(f) + pr
        coms = vc[:32 * m] * nb
        rng = hashlib.shake_256(b"pp%d%d" % (n, m)).digest(64 * nb)
        for ln in (0, 5, 41, 150):         # same message LENGTH for every proof: one position class, different sponge words
            states = b""
            for b_ in range(nb):
                st = transcript_new(b"app %d" % 7)
                st = transcript_append_message(st, b"session", bytes((b_ * 31 + q) & 0xff for q in range(ln)) if ln else b"")
                if ln == 0:
                    st = transcript_append_message(st, b"who", bytes([65 + b_]) * 4)
                states += st
            assert len({states[208 * b_ + 200:208 * b_ + 203] for b_ in range(nb)}) == 1 and len({states[208 * b_:208 * b_ + 200] for b_ in range(nb)}) == nb
            so, to = C.create_string_buffer(nb), C.create_string_buffer(208 * nb)
            rc = H.h_rp_transcript_compare_per_proof(n, m, nb, proofs, len(pr), coms, rng, states, so, to)
            assert rc == 0, (n, m, ln, rc)
            assert list(so.raw) == [0, 0, 0, 2, 0]
            assert to.raw[208 * 3:208 * 4] == states[208 * 3:208 * 4]
            for b_ in (0, 2, 4):
                _, _, ts_o = oracle.verify_ts(oracle.Gens(n, m), pr, coms[:32 * m], n, states[208 * b_:208 * (b_ + 1)], rng[64 * b_:64 * b_ + 64])
                assert ts_o == to.raw[208 * b_:208 * (b_ + 1)], (n, m, ln, b_)
        # states at different positions are refused by the harness (the queue never builds such a chain for the script)
        st_a = transcript_append_message(transcript_new(b"a"), b"x", b"12")
        st_b = transcript_append_message(transcript_new(b"a"), b"x", b"123")
        assert H.h_rp_transcript_compare_per_proof(n, m, 2, pr * 2, len(pr), coms[:64 * m], rng, st_a + st_b, so, to) == -2


def test_transcript_handed_back_as_of_the_reference_s_early_exit(H, oracle, golden):
    """validate_and_append_point returns Err BEFORE absorbing an identity A / S / T_1 / T_2 / L_i / R_i (transcript.rs:75-87; mod.rs:376-393,
    ipp.rs:217-222), so the caller's `&mut Transcript` stays at that message.  An identity encoding planted at every one of the 4 + 2k
    positions of every golden shape (and at two positions at once: the first one counts), from start states in several STROBE position
    classes: the byte-wise, the scripted and the 32-lane replay hand back the same state (the harness compares them lane by lane), and
    that state is the oracle's verify_ts state."""
    from bulletproofs_amd._lib import transcript_new, transcript_append_message
    label = golden["label"]
    vc = golden["vc_bytes"]
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        pr = bytes.fromhex(case["proof"])
        k = (n * m).bit_length() - 1
        offs = [0, 32, 64, 96] + [224 + 32 * j for j in range(2 * k)]   # transcript order: A, S, T_1, T_2, L_0, R_0, L_1, ...
        variants = []
        for u, o in enumerate(offs):
            z = bytearray(pr)
            z[o:o + 32] = bytes(32)
            variants.append(bytes(z))
        z = bytearray(pr)                      # two identities: the replay ends at the first
        z[offs[3]:offs[3] + 32] = bytes(32)
        z[offs[-1]:offs[-1] + 32] = bytes(32)
        variants.append(bytes(z))
        z = bytearray(pr)                      # an identity AND a non-canonical scalar: from_bytes fails first, transcript untouched
        z[offs[1]:offs[1] + 32] = bytes(32)
        z[160:192] = b"\xff" * 32
        variants.append(bytes(z))
        nb = len(variants)
        proofs = b"".join(variants)
        coms = vc[:32 * m] * nb
        rng = hashlib.shake_256(b"stop%d%d" % (n, m)).digest(64 * nb)
        gens = oracle.Gens(n, m)
        for pad in ([], [3], [77], [150], [166]):
            st = transcript_new(label)
            for j, ln in enumerate(pad):
                st = transcript_append_message(st, b"pad%d" % j, bytes((ln + q) & 0xff for q in range(ln)))
            so, to = C.create_string_buffer(nb), C.create_string_buffer(208 * nb)
            rc = H.h_rp_transcript_compare(n, m, nb, proofs, len(pr), coms, rng, st, 1, None, so, to)
            assert rc == 0, (n, m, pad, rc)
            for b_ in range(nb):
                rc_o, _, ts_o = oracle.verify_ts(gens, variants[b_], coms[:32 * m], n, st, rng[64 * b_:64 * b_ + 64])
                assert so.raw[b_] == rc_o, (n, m, pad, b_)
                assert ts_o == to.raw[208 * b_:208 * (b_ + 1)], (n, m, pad, b_)
            assert len({to.raw[208 * b_:208 * (b_ + 1)] for b_ in range(len(offs))}) == len(offs)   # every position leaves its own state
        # n m != 2^k: verification_scalars fails before the inner-product domain separator (ipp.rs:203-213); byte-wise replay
        if m >= 2:
            st = transcript_append_message(transcript_new(label), b"pad", b"1234567")
            z = bytearray(pr)
            z[64:96] = bytes(32)               # ... unless an identity T_1 ends the replay earlier
            for var in (pr, bytes(z)):
                so, to = C.create_string_buffer(1), C.create_string_buffer(208)
                assert H.h_rp_transcript_shape_stop(n, m // 2, var, len(pr), vc[:32 * (m // 2)], rng[:64], st, so, to) == 0
                rc_o, _, ts_o = oracle.verify_ts(oracle.Gens(n, m // 2), var, vc[:32 * (m // 2)], n, st, rng[:64])
                assert rc_o == 1 and so.raw[0] == 1 and ts_o == to.raw


def test_window_recoding_all_widths(H):
    """fb_recode / fb_nwin (msm_fixed.h): for every window width 2..20 the signed digits reconstruct the scalar and
    ceil(255 / W) windows suffice (W = 17 is the first width that saves a window: 15 instead of 16)."""
    L = T.L
    random.seed(5)
    vals = [0, 1, L - 1, 2**252, 2**253 - 1 if 2**253 - 1 < L else L - 2, (L - 1) // 2] + [random.randrange(L) for _ in range(40)]
    for W in range(2, 21):
        half = 1 << (W - 1)
        for v in vals:
            d = (C.c_uint32 * 130)()
            nwin = H.h_fb_recode(W, v.to_bytes(32, "little"), d)
            assert nwin == (255 + W - 1) // W
            assert all(0 <= d[i] < (1 << W) for i in range(nwin))
            assert sum((d[i] - half) << (W * i) for i in range(nwin)) == v, (W, hex(v))
    assert (255 + 16) // 17 == 15 and (255 + 15) // 16 == 16


def _rlc_expected(oracle, gg, n, m, label, proofs, plen, coms, rng, wts):
    """R = sum_i rho_i * MegaCheck_i over the proofs the front end accepts, by one oracle MSM over all weighted terms."""
    L = T.L
    all_s, all_p, included = b"", b"", []
    nb = len(proofs) // plen
    for b in range(nb):
        rc, sc_, pt_ = oracle.verify_terms(gg, proofs[plen * b:plen * (b + 1)], coms[32 * m * b:32 * m * (b + 1)], n, label, rng[64 * b:64 * b + 64])
        ok = rc == 0 and all(oracle.lib().oracle_point_decompress_ok(pt_[32 * j:32 * j + 32]) for j in range(len(pt_) // 32))
        included.append(ok)
        if not ok:
            continue
        rho = int.from_bytes(wts[64 * b:64 * b + 64], "little") % L
        for j in range(len(sc_) // 32):
            all_s += (int.from_bytes(sc_[32 * j:32 * j + 32], "little") * rho % L).to_bytes(32, "little")
        all_p += pt_
    st, enc = oracle.msm(all_s, all_p) if all_s else (0, bytes(32))
    assert st == 0
    return included, enc


@pytest.mark.parametrize("defer", [0, 1])
def test_batch_combination_pipeline_lane_by_lane(H, oracle, golden, defer):
    """rlc.h: limb accumulator, and the combined pipeline R = sum rho_i MegaCheck_i against one oracle MSM."""
    H.h_set_defer_emit(defer)          # (1: the weighted coefficients are parked by the leader and recoded by the group's lanes, rp_defer)
    L = T.L
    random.seed(11)
    for vals in ([1], [L - 1] * 3, [random.randrange(L) for _ in range(1000)], [L - 1] * 5000):
        o = C.create_string_buffer(32)
        H.h_rlc_sum(len(vals), b"".join(v.to_bytes(32, "little") for v in vals), o)
        assert int.from_bytes(o.raw, "little") == sum(vals) % L
    label = golden["label"]
    vc = golden["vc_bytes"]
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        if n * m > 32:
            continue
        pr = bytes.fromhex(case["proof"])
        gg = oracle.Gens(n, m)
        G2, H2, B2, Bb2 = gg.export()
        gens = Bb2 + B2 + G2 + H2
        bad = bytearray(pr)
        bad[128] ^= 1                 # t_x: parses, fails the check
        fmt = bytearray(pr)
        fmt[128:160] = b"\xff" * 32   # non-canonical scalar: FormatError, left out of the combination
        und = bytearray(pr)
        und[32] |= 1                  # S undecodable: VerificationError from the decoder, left out
        for name, batch in (("valid", [pr, pr, pr]), ("mixed", [pr, bytes(bad), bytes(fmt), bytes(und), pr])):
            nb = len(batch)
            proofs, coms = b"".join(batch), vc[:32 * m] * nb
            rng = hashlib.shake_256(b"r%d%d" % (n, m)).digest(64 * nb)
            wts = hashlib.shake_256(b"w%d%d" % (n, m)).digest(64 * nb)
            vd, bo = C.create_string_buffer(nb), C.create_string_buffer(33)
            assert H.h_rp_verify_rlc(4, 3, n, m, gens, n, m, nb, proofs, len(pr), coms, label, len(label), rng, wts, vd, bo) == 0
            included, enc = _rlc_expected(oracle, gg, n, m, label, proofs, len(pr), coms, rng, wts)
            assert bo.raw[1:33] == enc, (n, m, name)
            if name == "valid":
                assert bo.raw[0] == 0 and enc == bytes(32) and list(vd.raw) == [0, 0, 0]
            else:
                assert included == [True, True, False, False, True] and bo.raw[0] == 1 and enc != bytes(32)
                assert list(vd.raw) == [5, 5, 2, 1, 5]
    H.h_set_defer_emit(0)


def test_device_expanded_randomness_lane_by_lane(H, oracle, golden):
    """rp_shape::seed (rangeproof.h): where the caller brings no rng bytes / no combination weights, launch 1 expands them from one
    32-byte key per launch chain -- proof p's 64 bytes are block p of ChaCha20(key, nonce = domain).  (1) the block function shared by
    host and device code (csrc/chacha20.h) against the oracle's restatement (pinned on RFC 8439 / rand_chacha in test_oracle.py);
    (2) the per-proof pipeline without an rng buffer == the pipeline GIVEN those blocks, encodings and verdicts, contiguous and through
    a segment table whose odd items have no rng of their own; (3) the batch-combined pipeline without a weight buffer == the one
    given the weight blocks (the combined point of a failing batch depends on every weight)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "py"))
    from chacha_rng import chacha20_block
    key = hashlib.shake_256(b"chain-seed").digest(32)
    blk = C.create_string_buffer(64)
    for p_, dom in ((0, 1), (1, 1), (5, 2), (2**32 - 1, 2), (70000, 1)):
        H.h_chain_seed_block(key, p_, dom, blk)
        assert blk.raw == chacha20_block(key, p_, dom)
    label = golden["label"]
    vc = golden["vc_bytes"]
    case = [c for c in golden["cases"] if c["n"] == 8 and c["m"] == 2][0]
    n, m = 8, 2
    pr = bytes.fromhex(case["proof"])
    bad = bytearray(pr)
    bad[128] ^= 1
    fmt = bytearray(pr)
    fmt[128:160] = b"\xff" * 32
    batch = [pr, bytes(bad), pr, bytes(fmt), pr, pr, bytes(bad)]
    nb = len(batch)
    proofs, coms = b"".join(batch), vc[:32 * m] * nb
    gg = oracle.Gens(n, m)
    G2, H2, B2, Bb2 = gg.export()
    gens = Bb2 + B2 + G2 + H2
    rng_x = b"".join(chacha20_block(key, p_, 1) for p_ in range(nb))
    wts_x = b"".join(chacha20_block(key, p_, 2) for p_ in range(nb))
    other = hashlib.shake_256(b"unused").digest(64 * nb)
    # (2) per-proof pipeline
    exp = [oracle.verify(gg, batch[b], coms[32 * m * b:32 * m * (b + 1)], n, label, rng_x[64 * b:64 * b + 64]) for b in range(nb)]
    assert [e[0] for e in exp] == [0, 1, 0, 2, 0, 0, 1]
    for sizes in ([], [2, 1, 3, 1], [1] * 7):
        H.h_set_segments(len(sizes), (C.c_uint32 * max(len(sizes), 1))(*sizes))
        H.h_set_chain_seed(key, 1)
        vd, mo = C.create_string_buffer(nb), C.create_string_buffer(32 * nb)
        assert H.h_rp_verify(4, 3, n, m, gens, n, m, nb, proofs, len(pr), coms, label, len(label), other, vd, mo) == 0
        H.h_set_chain_seed(None, 0)
        for b in range(nb):
            assert vd.raw[b] == exp[b][0] and mo.raw[32 * b:32 * b + 32] == exp[b][1], (sizes, b)
    H.h_set_segments(0, (C.c_uint32 * 1)(0))
    # (3) batch-combined pipeline: weights from the key (rng given), then both from the key
    vd0, bo0 = C.create_string_buffer(nb), C.create_string_buffer(33)
    assert H.h_rp_verify_rlc(4, 3, n, m, gens, n, m, nb, proofs, len(pr), coms, label, len(label), rng_x, wts_x, vd0, bo0) == 0
    included, enc = _rlc_expected(oracle, gg, n, m, label, proofs, len(pr), coms, rng_x, wts_x)
    assert bo0.raw[0] == 1 and bo0.raw[1:33] == enc and enc != bytes(32) and list(vd0.raw) == [5, 5, 5, 2, 5, 5, 5]
    for flags, rng_arg in ((2, rng_x), (3, other)):
        H.h_set_chain_seed(key, flags)
        vd, bo = C.create_string_buffer(nb), C.create_string_buffer(33)
        assert H.h_rp_verify_rlc(4, 3, n, m, gens, n, m, nb, proofs, len(pr), coms, label, len(label), rng_arg, other, vd, bo) == 0
        H.h_set_chain_seed(None, 0)
        assert bo.raw == bo0.raw and vd.raw == vd0.raw, flags


def _linear_cases(oracle, n, tag):
    """five proofs of size n on the reference's test shape (linear_proof.rs:401-466): valid, a tampered, r non-canonical,
    an identity L_0 (or S undecodable at n = 1), wrong commitment"""
    insts = [oracle.linear_test_instance(n, b"%s-%d-%d" % (tag, n, j)) for j in range(5)]
    pl = len(insts[0]["proof"])
    bad = bytearray(insts[1]["proof"])
    bad[pl - 64] ^= 1                               # a tampered: still canonical with overwhelming probability
    insts[1] = dict(insts[1], proof=bytes(bad))
    fmt = bytearray(insts[2]["proof"])
    fmt[pl - 32:] = b"\xff" * 32                    # r >= l: FormatError (from_bytes, linear_proof.rs:383-384)
    insts[2] = dict(insts[2], proof=bytes(fmt))
    und = bytearray(insts[3]["proof"])
    if n > 1:
        und[0:32] = bytes(32)                       # L_0 = identity encoding: validate_and_append_point fails (:272)
    else:
        und[0] |= 1                                 # S: negative field element, does not decode (:217)
    insts[3] = dict(insts[3], proof=bytes(und))
    insts[4] = dict(insts[4], C=insts[0]["C"])      # somebody else's commitment
    return insts, pl


@pytest.mark.parametrize("n", [1, 2, 16, 64])
def test_linear_proof_front_end_lane_by_lane(H, oracle, n):
    """lin_prepare (public inputs, rounds, Gray-code subset products, <s, b>) + the variable-base pipeline against the
    oracle's restatement of LinearProof::verify (linear_proof.rs:175-236); sizes of the reference's tests (:470-487)."""
    insts, pl = _linear_cases(oracle, n, b"hlin")
    cat = lambda key: b"".join(i[key] for i in insts)
    nb = len(insts)
    vd, mo = C.create_string_buffer(nb), C.create_string_buffer(32 * nb)
    g0 = insts[0]
    assert H.h_lin_verify(n, nb, cat("proof"), pl, g0["label"], len(g0["label"]), cat("C"), g0["G"], g0["F"], g0["B"], cat("b"), 0, vd, mo) == 0
    st = oracle.transcript_new(g0["label"])
    for j, inst in enumerate(insts):
        rc, em = oracle.linear_verify(n, inst["proof"], st, inst["C"], inst["G"], inst["F"], inst["B"], inst["b"])
        assert vd.raw[j] == rc, (n, j, vd.raw[j], rc)
        if rc != 2 and not (j == 3):
            assert mo.raw[32 * j:32 * j + 32] == em, (n, j)
    assert list(vd.raw) == [0, 1, 2, 1, 1]
    # generator-table mode: the same verdicts and results when G, F, B are "the loaded generators" (B_blinding, B, G_0..)
    if n <= 16:
        vd3, mo3 = C.create_string_buffer(nb), C.create_string_buffer(32 * nb)
        gens = g0["B"] + g0["F"] + g0["G"]
        assert H.h_lin_verify_fixed(5, n, nb, cat("proof"), pl, g0["label"], len(g0["label"]), cat("C"), gens, cat("b"), 0, vd3, mo3) == 0
        assert vd3.raw == vd.raw
        for j in (0, 1, 4):
            assert mo3.raw[32 * j:32 * j + 32] == mo.raw[32 * j:32 * j + 32], (n, j)
    # one public vector shared by the batch; wrong n for the proof length
    vd2 = C.create_string_buffer(2)
    assert H.h_lin_verify(n, 2, g0["proof"] * 2, pl, g0["label"], len(g0["label"]), g0["C"] * 2, g0["G"], g0["F"], g0["B"], g0["b"], 1, vd2, None) == 0
    assert list(vd2.raw) == [0, 0]
    assert H.h_lin_verify(2 * n, 1, g0["proof"], pl, g0["label"], len(g0["label"]), g0["C"], g0["G"] * 2, g0["F"], g0["B"], g0["b"] * 2, 0, vd2, None) == 0
    assert vd2.raw[0] == 1 == oracle.linear_verify(2 * n, g0["proof"], st, g0["C"], g0["G"] * 2, g0["F"], g0["B"], g0["b"] * 2)[0]


@pytest.mark.parametrize("n", [1, 2, 8, 32])
def test_linear_prover_lane_by_lane(H, oracle, n):
    """the batched LinearProof prover (linear_prover.h: generators never folded, L_j / R_j / S over the original points)
    produces byte-identical proofs to the oracle's restatement of LinearProof::create (linear_proof.rs:40-173), leaves the
    transcript where the reference leaves it, and its proofs verify"""
    nb = 3
    insts = [oracle.linear_test_instance(n, b"hlinc-%d-%d" % (n, j)) for j in range(nb)]
    g0 = insts[0]
    st_app = oracle.transcript_append_message(oracle.transcript_new(b"prover app"), b"ctx", b"42")
    cat = lambda key: b"".join(i[key] for i in insts)
    pl = 32 * (2 * (n.bit_length() - 1) + 3)
    proofs, stb, tso = C.create_string_buffer(pl * nb), C.create_string_buffer(nb), C.create_string_buffer(208 * nb)
    assert H.h_lin_create(n, nb, st_app, cat("rng"), cat("C"), cat("r"), cat("a"), cat("b"), 0, g0["G"], g0["F"], g0["B"], proofs, stb, tso) == 0
    assert list(stb.raw) == [0] * nb
    for j, inst in enumerate(insts):
        rc, want = oracle.linear_create(n, st_app, inst["rng"], inst["C"], inst["r"], inst["a"], inst["b"], inst["G"], inst["F"], inst["B"])
        assert rc == 0 and proofs.raw[pl * j:pl * (j + 1)] == want, (n, j)
        assert oracle.linear_verify(n, want, st_app, inst["C"], inst["G"], inst["F"], inst["B"], inst["b"])[0] == 0
    # shared public vector; a non-canonical secret scalar is reported, not proved
    a_bad = bytearray(g0["a"] * 2)
    a_bad[32 * n:32 * n + 32] = b"\xff" * 32
    assert H.h_lin_create(n, 2, st_app, g0["rng"] * 2, g0["C"] * 2, g0["r"] * 2, bytes(a_bad), g0["b"], 1, g0["G"], g0["F"], g0["B"], proofs, stb, None) == 0
    assert stb.raw[0] == 0 and stb.raw[1] != 0 and proofs.raw[:pl] == oracle.linear_create(n, st_app, g0["rng"], g0["C"], g0["r"], g0["a"], g0["b"],
                                                                                          g0["G"], g0["F"], g0["B"])[1]


@pytest.mark.parametrize("n", [1, 2, 4, 32])
def test_standalone_ipp_front_end_lane_by_lane(H, oracle, n):
    """ipp_prepare (transcript, batch inversion, s_i products) + the variable-base pipeline against the oracle's
    restatement of InnerProductProof::verify (ipp.rs:260-326) on the reference's own test shape."""
    insts = [oracle.ipp_test_instance(n, b"innerproducttest", b"hipp-%d-%d" % (n, j)) for j in range(3)]
    pl = len(insts[0]["proof"])
    bad = bytearray(insts[1]["proof"])
    bad[-40] ^= 1                                   # a tampered: still canonical with overwhelming probability
    insts[1] = dict(insts[1], proof=bytes(bad))
    insts[2] = dict(insts[2], P=insts[2]["Q"])      # wrong P
    cat = lambda key: b"".join(i[key] for i in insts)
    vd, mo = C.create_string_buffer(3), C.create_string_buffer(96)
    assert H.h_ipp_verify(n, 3, cat("proof"), pl, b"innerproducttest", 16, cat("Gf"), cat("Hf"), cat("P"), cat("Q"), cat("G"), cat("H"), vd, mo) == 0
    for j, inst in enumerate(insts):
        rc, em = oracle.ipp_verify(n, inst["proof"], b"innerproducttest", inst["Gf"], inst["Hf"], inst["P"], inst["Q"], inst["G"], inst["H"])
        assert vd.raw[j] == rc and mo.raw[32 * j:32 * j + 32] == em, (n, j)
    assert list(vd.raw) == [0, 1, 1]


@pytest.mark.parametrize("n", [2, 4, 32, 64])
def test_verification_scalars_front_end_and_its_transcript_on_every_path(H, oracle, n):
    """ipp_vs_front_thread / ipp_vs_s_thread (bpgpu_ipp_verification_scalars; ipp.rs:198-253): u_i^2, u_i^-2, s_i and the caller's transcript
    against the oracle (pinned on the Python twin, tests/test_oracle.py) -- valid proofs, an identity encoding at every L_i / R_i position (the
    state as of that message: domain separator and earlier rounds in, transcript.rs:75-87), per-proof start states and one for the batch."""
    from bulletproofs_amd._lib import transcript_new, transcript_append_message
    label = b"innerproducttest"
    k = n.bit_length() - 1
    pr = oracle.ipp_test_instance(n, label, b"hvs%d" % n)["proof"]
    pl = len(pr)
    variants = [pr] + [pr[:32 * u] + bytes(32) + pr[32 * u + 32:] for u in range(2 * k)]
    nb = len(variants)
    proofs = b"".join(variants)
    shared = transcript_append_message(transcript_new(label), b"earlier", b"message of the parent protocol")
    per = b"".join(transcript_append_message(transcript_new(label), b"p", bytes(j for j in range(3 * i + 1))) for i in range(nb))
    for per_proof, states in ((0, shared), (1, per)):
        so, us, ui = C.create_string_buffer(nb), C.create_string_buffer(32 * k * nb), C.create_string_buffer(32 * k * nb)
        sv, to = C.create_string_buffer(32 * n * nb), C.create_string_buffer(208 * nb)
        assert H.h_ipp_vs(n, nb, proofs, pl, states, per_proof, so, us, ui, sv, to) == 0
        for i in range(nb):
            start = states[208 * i:208 * (i + 1)] if per_proof else shared
            rc, eus, eui, es, est = oracle.ipp_verification_scalars(n, variants[i], start)
            assert so.raw[i] == rc == (0 if i == 0 else 1), (n, per_proof, i)
            assert to.raw[208 * i:208 * (i + 1)] == est, (n, per_proof, i)
            if rc == 0:
                assert us.raw[32 * k * i:32 * k * (i + 1)] == eus and ui.raw[32 * k * i:32 * k * (i + 1)] == eui and sv.raw[32 * n * i:32 * n * (i + 1)] == es
        assert len({to.raw[208 * i:208 * (i + 1)] for i in range(nb)}) == nb


def test_util_rs_scalar_tests_on_the_device_code(H, oracle):
    """src/util.rs:274-351 -- exp_2_is_powers_of_2, test_inner_product (<a, b> = 40), test_scalar_exp (the fixed scalar),
    test_sum_of_powers (x = 10, n in {1, .., 64}) -- against the device's scalar arithmetic (sc25519.h, rangeproof.h) compiled
    for the host, and the oracle's; the reference's values are small enough to state as integers."""
    ell = 2 ** 252 + 27742317777372353535851937790883648493
    le = lambda v: (v % ell).to_bytes(32, "little")

    def op(code, a, b=0):
        o = C.create_string_buffer(32)
        H.h_sc_op(code, le(a), le(b), o)
        return int.from_bytes(o.raw, "little")
    # exp_iter(2): 1, 2, 4, 8
    acc, seen = 1, []
    for _ in range(4):
        seen.append(acc)
        acc = op(0, acc, 2)
    assert seen == [1, 2, 4, 8]
    # inner_product([1, 2, 3, 4], [2, 3, 4, 5]) = 40
    ip = 0
    for a, b in zip((1, 2, 3, 4), (2, 3, 4, 5)):
        ip = op(1, ip, op(0, a, b))
    assert ip == 40
    # scalar_exp_vartime on the reference's fixed scalar: repeated products against integer powers mod l
    x = int.from_bytes(b"\x84\xfc\xbcOx\x12\xa0\x06\xd7\x91\xd9z:'\xdd\x1e!CE\xf7\xb1\xb9Vz\x810sD\x96\x85\xb5\x07", "little")
    assert x < ell
    acc = 1
    for e in range(1, 65):
        acc = op(0, acc, x)
        if e in (1, 2, 3, 4, 5, 64):
            assert acc == pow(x, e, ell), e
    o = C.create_string_buffer(32)
    oracle.lib().oracle_scalar_mul(le(x), le(x), o)
    assert int.from_bytes(o.raw, "little") == pow(x, 2, ell)
    assert op(0, x, op(4, x)) == 1 and op(7, x) == pow(x, ell - 2, ell) == op(8, x)         # the three inversions agree
    # sum_of_powers(10, n): 1, 11, 1111, 11111111, ... and the closed form for the larger ones
    for lg in range(0, 7):
        H.h_sum_of_powers_pow2(le(10), lg, o)
        n = 1 << lg
        assert int.from_bytes(o.raw, "little") == sum(pow(10, i, ell) for i in range(n)) % ell == (int("1" * n) % ell), n


def _share_cases(oracle, g, n, m, tag):
    """an honest aggregation of m parties and what its dealer would audit: returns (party indices, shares, bit commitments, poly
    commitments, challenges, expected verdicts) with a few dishonest variations appended"""
    vals = [((1 << n) - 1 - 977 * i) % (1 << n) for i in range(m)]
    bl = b"".join(hashlib.shake_256(b"%s-bl%d" % (tag, i)).digest(31) + b"\x00" for i in range(m))
    r = oracle.prove_shares(g, vals, bl, n, b"mpc audit", tag)
    sl = 32 * (3 + 2 * n)
    S = [r["shares"][sl * j:sl * (j + 1)] for j in range(m)]
    BC = [r["bit_commitments"][96 * j:96 * j + 96] for j in range(m)]
    PC = [r["poly_commitments"][64 * j:64 * j + 64] for j in range(m)]
    idx, sh, bc, pc = list(range(m)), list(S), list(BC), list(PC)

    def add(j, s_, b_, p_):
        idx.append(j); sh.append(bytes(s_)); bc.append(bytes(b_)); pc.append(bytes(p_))
    t = bytearray(S[0]); t[96 + 7] ^= 1; add(0, t, BC[0], PC[0])                       # l_vec tampered: t_x != <l, r>
    t = bytearray(S[0]); t[40] ^= 1; add(0, t, BC[0], PC[0])                           # t_x_blinding tampered: t_check fails
    t = bytearray(S[0]); t[70] ^= 1; add(0, t, BC[0], PC[0])                           # e_blinding tampered: P_check fails
    add(0, S[0], BC[0], PC[0][32:] + PC[0][:32])                                       # T_1 and T_2 swapped
    add(0, S[0], BC[0][:32] + BC[0][64:] + BC[0][32:64], PC[0])                        # A_j and S_j swapped
    t = bytearray(S[0]); t[0:32] = b"\xff" * 32; add(0, t, BC[0], PC[0])               # t_x not canonical
    if m > 1:
        add(1, S[0], BC[0], PC[0])                                                     # party 0's share audited as party 1
    add(m + 50, S[0], BC[0], PC[0])                                                    # j >= party_capacity (check_size)
    return idx, sh, bc, pc, r["challenges"], [0] * m + [1] * (len(idx) - m)


@pytest.mark.parametrize("n,m", [(8, 1), (16, 4), (64, 2)])
def test_share_audit_lane_by_lane(H, oracle, n, m):
    """audit.h (ProofShare::audit_share, messages.rs:85-167) against the oracle's restatement: verdicts and both check points,
    for the honest shares of an m-party aggregation and dishonest variations of them"""
    cap, parties = 64, 4
    g = oracle.Gens(cap, parties)
    Gc, Hc, Bp, Bb = g.export()
    gens = Bb + Bp + Gc + Hc
    idx, sh, bc, pc, chal, expect = _share_cases(oracle, g, n, m, b"haud-%d-%d" % (n, m))
    ns = len(idx)
    vd, chk = C.create_string_buffer(ns), C.create_string_buffer(64 * ns)
    pi = (C.c_uint32 * ns)(*idx)
    assert H.h_audit_shares(n, ns, cap, parties, gens, pi, b"".join(sh), b"".join(bc), b"".join(pc), chal, 1, vd, chk) == 0
    assert list(vd.raw) == expect
    for k in range(ns):
        rc, out = oracle.audit_share(g, n, idx[k], sh[k], bc[k], pc[k], chal)
        assert rc == expect[k], (k, rc)
        if out[:32] != b"\xff" * 32:
            assert chk.raw[64 * k:64 * k + 32] == out[:32], k
        if out[32:] != b"\xff" * 32:
            assert chk.raw[64 * k + 32:64 * k + 64] == out[32:], k


def test_keccak_on_25_lanes_equals_the_serial_permutation_and_sha3(H):
    """keccak.h: Keccak-f[1600] with one 64-bit state word per lane of a 32-lane group (option transcript_coop: narrow chains replay
    their transcripts 32 lanes per proof) -- four exchange phases per round, each run here on a snapshot of the 25 lanes' values with
    the kernels' own phase functions (rho offsets, the pi source table, neighbour indices).  == the serial permutation on random
    states with and without a 42-word XOR mask; the serial one is pinned on SHA3 / SHAKE vectors elsewhere in this file, and on the
    all-zero state's published first output lane here."""
    z_c, z_s = C.create_string_buffer(200), C.create_string_buffer(200)
    H.h_keccak_coop(bytes(200), None, 0, z_c, z_s)
    assert z_c.raw == z_s.raw and z_c.raw[:8] == bytes.fromhex("e7dde140798f25f1")      # Keccak-f[1600](0): first lane F1258F7940E1DDE7
    for i in range(40):
        st = hashlib.shake_256(b"kc%d" % i).digest(200)
        mask = (C.c_uint32 * 42)(*[int.from_bytes(hashlib.shake_256(b"km%d-%d" % (i, q)).digest(4), "little") for q in range(42)])
        for nm, mk in ((0, None), (42, mask), (7, mask)):
            a, b = C.create_string_buffer(200), C.create_string_buffer(200)
            H.h_keccak_coop(st, mk, nm, a, b)
            assert a.raw == b.raw and a.raw != st, (i, nm)

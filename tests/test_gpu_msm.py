"""GPU parity tests for the multiscalar-multiplication entry points (C ABI:
bpgpu_msm_batch / bpgpu_msm_batch_shared) against the CPU oracle, on the same
seeded inputs; bit-exact on the 32-byte ristretto encoding of every result."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


def _scalar(tag):
    L = 2**252 + 27742317777372353535851937790883648493
    return (int.from_bytes(hashlib.shake_256(tag).digest(64), "little") % L).to_bytes(32, "little")


def _points(oracle, seed, n):
    import ctypes as C
    out = C.create_string_buffer(32)
    pts = []
    for i in range(n):
        oracle.lib().oracle_from_uniform_bytes(hashlib.shake_256(b"%s-p%d" % (seed, i)).digest(64), out)
        pts.append(out.raw)
    return b"".join(pts)


def _rand_msm(oracle, seed, n):
    return b"".join(_scalar(b"%s-s%d" % (seed, i)) for i in range(n)), _points(oracle, seed, n)


@pytest.fixture(scope="module")
def ctx():
    import bulletproofs_amd as bp
    c = bp.Context(0)
    yield c
    c.close()


def test_msm_batch_random_ragged(ctx, oracle):
    sizes = [0, 1, 2, 31, 32, 33, 64, 147, 200, 1, 0, 700]
    S, P = b"", b""
    for k, n in enumerate(sizes):
        s, p = _rand_msm(oracle, b"m%d" % k, n)
        S += s
        P += p
    out, st = ctx.msm_batch(sizes, S, P)
    off = 0
    for k, n in enumerate(sizes):
        exp = oracle.msm(S[off:off + 32 * n], P[off:off + 32 * n])
        off += 32 * n
        assert st[k] == 0 and out[32 * k:32 * k + 32] == exp[1], (k, n)


def test_msm_batch_edge_scalars_and_bad_inputs(ctx, oracle):
    L = 2**252 + 27742317777372353535851937790883648493
    sp = [0, 1, L - 1, 8, int("8" * 63, 16) % L, 2**252, 7, 2**128]
    s = b"".join(x.to_bytes(32, "little") for x in sp)
    p = _points(oracle, b"edge", len(sp))
    out, st = ctx.msm_batch([len(sp)], s, p)
    assert st[0] == 0 and out == oracle.msm(s, p)[1]
    # P + (-1)P = identity -> all-zero encoding
    two_s = (1).to_bytes(32, "little") + (L - 1).to_bytes(32, "little")
    out, st = ctx.msm_batch([2], two_s, p[:32] + p[:32])
    assert st[0] == 0 and out == bytes(32)
    # undecodable point -> status 1 (Option::None), zero output; other MSMs of the batch unaffected
    bad = bytearray(p)
    bad[0] |= 1
    out, st = ctx.msm_batch([len(sp), len(sp)], s + s, bytes(bad) + p)
    assert st[0] == 1 and out[:32] == bytes(32)
    assert st[1] == 0 and out[32:] == oracle.msm(s, p)[1]
    # non-canonical scalar -> status 2
    s2 = bytearray(s)
    s2[0:32] = L.to_bytes(32, "little")
    out, st = ctx.msm_batch([len(sp)], bytes(s2), p)
    assert st[0] == 2


def test_golden_proofs_mega_check_is_identity_on_gpu(ctx, oracle, oracle_gens_64_8, golden):
    """The 16 reference proofs (tests/range_proof.rs:16-95): the oracle expands the MSM terms
    (mod.rs:421-443), the GPU evaluates them; every result must be the identity encoding."""
    rng64 = hashlib.shake_256(b"golden-gpu").digest(64)
    sizes, S, P = [], b"", b""
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        rc, sc, pt = oracle.verify_terms(oracle_gens_64_8, bytes.fromhex(case["proof"]), golden["vc_bytes"][:32 * m], n,
                                         golden["label"], rng64)
        assert rc == 0
        sizes.append(len(sc) // 32)
        S += sc
        P += pt
    assert sizes == [29, 48, 84, 154, 47, 82, 150, 284, 81, 148, 280, 542, 147, 278, 538, 1056]
    out, st = ctx.msm_batch(sizes, S, P)
    assert st == bytes(16) and out == bytes(32 * 16)


@pytest.mark.parametrize("W", [8, 5])
def test_msm_batch_shared_matches_general_path_and_oracle(oracle, W):
    import bulletproofs_amd as bp
    c = bp.Context(0, fixed_window_bits=W)
    g = oracle.Gens(16, 2)
    G, H, B, Bb = g.export()
    c.gens_load(16, 2, G, H, B, Bb)
    n, m, nb, nu = 16, 2, 70, 9
    ngen = 2 * n * m + 2
    gen_pts = Bb + B + G[:32 * 16] + G[32 * 16:32 * 32] + H[:32 * 16] + H[32 * 16:32 * 32]
    GS, US, UP = b"", b"", b""
    for b in range(nb):
        GS += b"".join(_scalar(b"g%d-%d" % (b, i)) for i in range(ngen))
        s, p = _rand_msm(oracle, b"u%d" % b, nu)
        US += s
        UP += p
    out, st = c.msm_batch_shared(n, m, nb, nu, GS, US, UP)
    for b in range(nb):
        scs = GS[32 * ngen * b:32 * ngen * (b + 1)] + US[32 * nu * b:32 * nu * (b + 1)]
        pts = gen_pts + UP[32 * nu * b:32 * nu * (b + 1)]
        exp = oracle.msm(scs, pts)
        assert st[b] == 0 and out[32 * b:32 * b + 32] == exp[1], b
    # smaller (n, m) than the loaded capacity selects the right generator subset (generators.rs:207-259)
    n2, m2 = 8, 1
    ngen2 = 2 * n2 * m2 + 2
    gs = b"".join(_scalar(b"h%d" % i) for i in range(ngen2))
    out, st = c.msm_batch_shared(n2, m2, 1, 0, gs, b"", b"")
    pts = Bb + B + G[:32 * 8] + H[:32 * 8]
    assert st[0] == 0 and out == oracle.msm(gs, pts)[1]
    c.close()


def test_r1cs_shape_msm_config5(oracle):
    """BASELINE config 5 shape (r1cs/verifier.rs:459-491, k=1024 shuffle): one MSM of 6179 terms =
    4098 generator terms (BulletproofGens::new(2048, 1) + Pedersen) + 2081 per-proof points; the reference
    has no golden vector for it (feature disabled), so parity is GPU == oracle on seeded inputs.
    Evaluated through both entry points."""
    import bulletproofs_amd as bp
    c = bp.Context(0)
    c.gens_create(2048, 1)
    g = oracle.Gens(2048, 1)
    G, H, B, Bb = g.export()
    assert c.gens_export() == (G, H, B, Bb)
    n, m, nu, nb = 2048, 1, 2081, 2
    ngen = 2 * n * m + 2
    gen_pts = Bb + B + G + H
    GS = b"".join(_scalar(b"r1cs-g%d" % i) for i in range(ngen * nb))
    UP = _points(oracle, b"r1cs-u", nu) * nb
    US = b"".join(_scalar(b"r1cs-u%d" % i) for i in range(nu * nb))
    out, st = c.msm_batch_shared(n, m, nb, nu, GS, US, UP)
    flat_s, flat_p = b"", b""
    for b in range(nb):
        scs = GS[32 * ngen * b:32 * ngen * (b + 1)] + US[32 * nu * b:32 * nu * (b + 1)]
        pts = gen_pts + UP[32 * nu * b:32 * nu * (b + 1)]
        assert len(scs) // 32 == 6179
        exp = oracle.msm(scs, pts)
        assert st[b] == 0 and out[32 * b:32 * b + 32] == exp[1], b
        flat_s += scs
        flat_p += pts
    out2, st2 = c.msm_batch([6179, 6179], flat_s, flat_p)
    assert st2 == bytes(2) and out2 == out
    c.close()


def test_device_pointer_entry_points_with_torch_streams(oracle):
    """The *_dev twins: device pointers (torch tensors) + a caller-owned HIP stream; results stay on the device
    and are ordered on that stream."""
    import ctypes as C
    import torch
    import bulletproofs_amd as bp
    L = bp.lib()
    dev = torch.device("cuda", 0)
    c = bp.Context(0, fixed_window_bits=8)
    g = oracle.Gens(8, 1)
    G, H, B, Bb = g.export()
    c.gens_load(8, 1, G, H, B, Bb)
    n, m, nb, nu = 8, 1, 5, 3
    ngen = 2 * n * m + 2
    GS = b"".join(_scalar(b"dg%d" % i) for i in range(ngen * nb))
    US = b"".join(_scalar(b"du%d" % i) for i in range(nu * nb))
    UP = _points(oracle, b"dp", nu * nb)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_gs, d_us, d_up = to_dev(GS), to_dev(US), to_dev(UP)
    d_out = torch.zeros(32 * nb, dtype=torch.uint8, device=dev)
    d_st = torch.full((nb,), 9, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    rc = L.bpgpu_msm_batch_shared_dev(c.h, n, m, nb, nu, d_gs.data_ptr(), d_us.data_ptr(), d_up.data_ptr(), d_out.data_ptr(),
                                      d_st.data_ptr(), s.cuda_stream)
    assert rc == 0, L.bpgpu_last_error(c.h)
    # general entry point on the same stream, ragged
    flat_s = GS[:32 * ngen] + US[:32 * nu]
    flat_p = Bb + B + G + H + UP[:32 * nu]
    nt = (C.c_uint32 * 2)(ngen + nu, 2)
    d_s2, d_p2 = to_dev(flat_s + US[:64]), to_dev(flat_p + UP[:64])
    d_out2 = torch.zeros(64, dtype=torch.uint8, device=dev)
    d_st2 = torch.full((2,), 9, dtype=torch.uint8, device=dev)
    rc = L.bpgpu_msm_batch_dev(c.h, 2, nt, d_s2.data_ptr(), d_p2.data_ptr(), d_out2.data_ptr(), d_st2.data_ptr(), s.cuda_stream)
    assert rc == 0, L.bpgpu_last_error(c.h)
    s.synchronize()
    out, st = bytes(d_out.cpu().numpy()), bytes(d_st.cpu().numpy())
    gp = Bb + B + G + H
    for b in range(nb):
        exp = oracle.msm(GS[32 * ngen * b:32 * ngen * (b + 1)] + US[32 * nu * b:32 * nu * (b + 1)], gp + UP[32 * nu * b:32 * nu * (b + 1)])
        assert st[b] == 0 and out[32 * b:32 * b + 32] == exp[1]
    out2 = bytes(d_out2.cpu().numpy())
    assert bytes(d_st2.cpu().numpy()) == bytes(2)
    assert out2[:32] == out[:32] and out2[32:] == oracle.msm(US[:64], UP[:64])[1]
    # misaligned device pointer is rejected by the range-proof entry point (bpgpu.h: 4-byte alignment)
    d_misc = torch.zeros(4096, dtype=torch.uint8, device=dev)
    rc = L.bpgpu_rangeproof_verify_batch_dev(c.h, 8, 1, 1, d_misc.data_ptr() + 1, 480, d_misc.data_ptr(), b"x", 1, None, d_st.data_ptr(), None, None)
    assert rc == -1
    c.close()


def _fast_points(oracle, seed, n, distinct=257):
    """n encodings drawn (with repetition) from `distinct` derived points: large MSMs without n Elligator calls in Python"""
    base = _points(oracle, seed, min(n, distinct))
    k = len(base) // 32
    return b"".join(base[32 * ((7 * i + i // k) % k):32 * ((7 * i + i // k) % k) + 32] for i in range(n))


def _fast_scalars(seed, n):
    raw = bytearray(hashlib.shake_256(seed).digest(32 * n))
    for i in range(31, len(raw), 32):
        raw[i] &= 0x0f            # < 2^252 < l: canonical
    return bytes(raw)


@pytest.mark.parametrize("sizes", [[256], [2081], [6179], [20000], [40000], [0, 300, 192, 1000, 5, 191, 2081], [7000, 0, 6500]])
def test_bucket_path_sizes_bit_exact(ctx, oracle, sizes):
    """The bucket (Pippenger) path of bpgpu_msm_batch (bucket.h; forced here with bucket_min_terms = 1, by default taken
    from 1536 terms per MSM on average; c = 8 below 6000 terms, c = 12 above): sizes around the thresholds, the R1CS verifier's 2081 / 6179 (r1cs/verifier.rs:459-491), 20 000,
    ragged batches with empty MSMs -- each result bit-exact vs the oracle's MSM (reference split Straus / Pippenger), and
    equal to the table-lookup path on the same inputs."""
    import bulletproofs_amd as bp
    S = b"".join(_fast_scalars(b"bk-%d-%d" % (k, n), n) for k, n in enumerate(sizes))
    P = b"".join(_fast_points(oracle, b"bk-%d" % k, n) for k, n in enumerate(sizes))
    cb = bp.Context(0)
    cb.set_option("bucket_min_terms", 1)                # force the bucket path
    out, st = cb.msm_batch(sizes, S, P)
    off = 0
    for k, n in enumerate(sizes):
        exp = oracle.msm(S[off:off + 32 * n], P[off:off + 32 * n])
        off += 32 * n
        assert st[k] == 0 and out[32 * k:32 * k + 32] == exp[1], (k, n)
    c2 = bp.Context(0)
    c2.set_option("bucket_min_terms", 2**31 - 1)       # force the table-lookup path
    out2, st2 = c2.msm_batch(sizes, S, P)
    c2.close()
    assert out2 == out and st2 == st
    assert ctx.msm_batch(sizes, S, P) == (out, st)      # and whatever the default picks
    if sizes == [2081]:      # a bad point / a non-canonical scalar inside a bucket-path MSM
        bad = bytearray(P)
        bad[32 * 1000] |= 1
        o, s_ = cb.msm_batch(sizes, S, bytes(bad))
        assert s_[0] == 1 and o == bytes(32)
        L = 2**252 + 27742317777372353535851937790883648493
        s2 = bytearray(S)
        s2[32 * 77:32 * 78] = L.to_bytes(32, "little")
        o, s_ = cb.msm_batch(sizes, bytes(s2), P)
        assert s_[0] == 2
    cb.close()


@pytest.mark.parametrize("n,kind", [(3000, "equal"), (8192, "ones"), (20000, "equal"), (20000, "short128"), (7000, "three-values")])
def test_bucket_path_structured_scalars_take_the_heavy_pass(ctx, oracle, n, kind):
    """Scalars that bk_recode's stirring cannot spread (equal, tiny, short ones that share their upper windows): whole windows'
    worth of terms land in ONE bucket.  A lane adds at most four times the average population of its bucket; the rest goes through
    k_bk_heavy, 64 lanes at a time (bucket.h stage 3b; lane by lane in the CPU harness).  Bit-exact vs the oracle's MSM and vs the
    table-lookup path."""
    import bulletproofs_amd as bp
    L = 2**252 + 27742317777372353535851937790883648493
    if kind == "equal":
        S = _scalar(b"hv-eq-%d" % n) * n
    elif kind == "ones":
        S = (1).to_bytes(32, "little") * n
    elif kind == "short128":
        S = b"".join(hashlib.shake_256(b"hv-sh-%d" % i).digest(16) + bytes(16) for i in range(n))
    else:
        vals = [_scalar(b"hv-3-%d" % j) for j in range(3)]
        S = b"".join(vals[i % 3] for i in range(n))
    P = _fast_points(oracle, b"hv-%s" % kind.encode(), n)
    exp = oracle.msm(S, P)[1]
    cb = bp.Context(0)
    cb.set_option("bucket_min_terms", 1)
    out, st = cb.msm_batch([n], S, P)
    cb.close()
    assert st[0] == 0 and out == exp
    assert ctx.msm_batch([n], S, P) == (out, st)
    # a batch in which one MSM is crowded and its neighbours are not
    S2 = _fast_scalars(b"hv-nb", 2000) + S[:32 * 2500] + _fast_scalars(b"hv-nc", 1800)
    P2 = _fast_points(oracle, b"hv-nb", 2000) + P[:32 * 2500] + _fast_points(oracle, b"hv-nc", 1800)
    cb = bp.Context(0)
    cb.set_option("bucket_min_terms", 1)
    out, st = cb.msm_batch([2000, 2500, 1800], S2, P2)
    cb.close()
    exp3 = [oracle.msm(S2[32 * a:32 * b], P2[32 * a:32 * b])[1] for a, b in ((0, 2000), (2000, 4500), (4500, 6300))]
    assert st == bytes(3) and out == b"".join(exp3)


@pytest.mark.parametrize("nb", [1, 3, 50])
def test_fused_bucket_chain_options_all_give_the_oracle_encoding(oracle, nb):
    """Every option of the fused bucket chain (include/bpgpu.h: bucket_chain, bucket_lanes, bucket_fast_tail, fb_walk_waves) flipped on the
    config-5 shape at three batch widths (one MSM: 256-lane window workgroups + the short-chain tail; 3: 128 lanes; 50: one wavefront per
    (MSM, window) + the one-lane-per-window tail): the same encodings as the oracle's, whatever the decomposition."""
    import bulletproofs_amd as bp
    c = bp.Context(0, fixed_window_bits=6)
    c.gens_create(2048, 1)
    g = oracle.Gens(2048, 1)
    G, H, B, Bb = g.export()
    n, m, nu = 2048, 1, 700
    ngen = 2 * n * m + 2
    gen_pts = Bb + B + G + H
    pts1 = _points(oracle, b"opt-u", nu)
    GS = b"".join(_scalar(b"opt-g%d" % (i % (2 * ngen))) for i in range(ngen * nb))
    US = b"".join(_scalar(b"opt-u%d" % (i % (3 * nu))) for i in range(nu * nb))
    UP = pts1 * nb
    c.set_option("bucket_min_terms", 1)
    expect = b""
    for b in range(min(nb, 3)):     # (the inputs repeat with period 2 / 3 MSMs: three oracle MSMs cover every batch width)
        scs = GS[32 * ngen * b:32 * ngen * (b + 1)] + US[32 * nu * b:32 * nu * (b + 1)]
        expect += oracle.msm(scs, gen_pts + pts1)[1]
    base, _ = c.msm_batch_shared(n, m, nb, nu, GS, US, UP)
    assert base[:len(expect)] == expect
    for key, values in (("bucket_chain", (1,)), ("bucket_lanes", (64, 128, 256)), ("bucket_fast_tail", (0, 1)), ("fb_walk_waves", (64, 4096))):
        before = c.get_option(key)
        for v in values:
            c.set_option(key, v)
            assert c.get_option(key) == v
            out, st = c.msm_batch_shared(n, m, nb, nu, GS, US, UP)
            assert st == bytes(nb) and out == base, (key, v)
        c.set_option(key, before if key != "fb_walk_waves" else 0)
    c.close()


def test_few_small_msms_take_the_narrow_form_and_agree_with_the_batch_form_and_the_oracle(oracle):
    """Option msm_narrow (round 6; csrc/k_msm.hip k_vb_prepare_hi / k_vb_window_hi / k_vb_tail_narrow): at most 16 MSMs of at most 768 terms
    in all -- optional_multiscalar_mul called from one thread with a Straus-size MSM (src/range_proof/mod.rs:421, inner_product_proof.rs:308) --
    run with second tables of the points' 2^128 multiples and a 32-window chain.  Same encodings and status bytes as the batch form and the
    oracle: the reference's golden MSM sizes, ragged batches with empty MSMs, edge scalars (hi or lo half zero, l - 1), undecodable points,
    non-canonical scalars."""
    import bulletproofs_amd as bp
    a, b = bp.Context(0), bp.Context(0)
    a.set_option("msm_narrow", 0)
    b.set_option("msm_narrow", 1)
    L = 2**252 + 27742317777372353535851937790883648493
    try:
        for k, sizes in enumerate([[147], [29], [1], [768], [1024], [81, 0, 148], [48] * 16, [0], [0, 0, 3], [542, 220], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]]):
            S, P = b"", b""
            for j, n in enumerate(sizes):
                s, p = _rand_msm(oracle, b"nm%d-%d" % (k, j), n)
                S += s
                P += p
            ra, rb = a.msm_batch(sizes, S, P), b.msm_batch(sizes, S, P)
            assert ra == rb, sizes
            off = 0
            for j, n in enumerate(sizes):
                exp = oracle.msm(S[off:off + 32 * n], P[off:off + 32 * n])
                off += 32 * n
                assert rb[1][j] == 0 and rb[0][32 * j:32 * j + 32] == exp[1], (sizes, j)
        sp = [0, 1, L - 1, 8, int("8" * 63, 16) % L, 2**252, 7, 2**128, 2**128 - 1, (2**124 - 1) << 128, 2**252 + 1]
        s = b"".join(x.to_bytes(32, "little") for x in sp)
        p = _points(oracle, b"nedge", len(sp))
        ra, rb = a.msm_batch([len(sp)], s, p), b.msm_batch([len(sp)], s, p)
        assert ra == rb and rb[1][0] == 0 and rb[0] == oracle.msm(s, p)[1]
        # P - P: the identity's all-zero encoding; then an undecodable point and a non-canonical scalar: status bytes 1 / 2, zero encodings
        one, m1 = (1).to_bytes(32, "little"), (L - 1).to_bytes(32, "little")
        assert b.msm_batch([2], one + m1, p[:32] * 2) == (bytes(32), bytes(1))
        bad_pt = bytes([p[0] | 1]) + p[1:32]
        sizes = [3, 2, 4]
        S = s[:32 * 9]
        P = p[:64] + bad_pt + p[96:32 * 9]
        S2 = S[:32 * 3] + (L).to_bytes(32, "little") + S[32 * 4:]
        for SS, PP in ((S, P), (S2, p[:32 * 9])):
            ra, rb = a.msm_batch(sizes, SS, PP), b.msm_batch(sizes, SS, PP)
            assert ra == rb and list(rb[1]).count(0) == 2
    finally:
        a.close()
        b.close()


def test_msm_batch_shared_narrow_form_vs_batch_form_and_oracle(oracle):
    """bpgpu_msm_batch_shared with a few MSMs of a few own points -- the mega-check of ONE 64-bit proof as the caller's own MSM call (130 generator
    terms from the tables + 17 points; src/range_proof/mod.rs:421-445) -- in the narrow form (option msm_narrow: k_vb_prepare_hi, k_vb_window_hi,
    the generator half as one launch, k_shared_tail_narrow) against the batch form and the oracle; the real terms of a golden-shape proof included."""
    import bulletproofs_amd as bp
    a, b = bp.Context(0, fixed_window_bits=10), bp.Context(0, fixed_window_bits=10)
    a.set_option("msm_narrow", 0)
    b.set_option("msm_narrow", 1)
    g = oracle.Gens(64, 1)
    G, H, B, Bb = g.export()
    for c in (a, b):
        c.gens_load(64, 1, G, H, B, Bb)
    n, m = 64, 1
    ngen = 2 * n * m + 2
    gen_pts = Bb + B + G + H
    try:
        for nb, nu in ((1, 17), (1, 1), (3, 40), (16, 48), (2, 300), (1, 768), (1, 769)):
            GS, US, UP = b"", b"", b""
            for i in range(nb):
                GS += b"".join(_scalar(b"sg%d-%d-%d" % (nb, i, j)) for j in range(ngen))
                s, p = _rand_msm(oracle, b"su%d-%d-%d" % (nb, nu, i), nu)
                US += s
                UP += p
            ra, rb = a.msm_batch_shared(n, m, nb, nu, GS, US, UP), b.msm_batch_shared(n, m, nb, nu, GS, US, UP)
            assert ra == rb, (nb, nu)
            for i in range(min(nb, 3)):
                exp = oracle.msm(GS[32 * ngen * i:32 * ngen * (i + 1)] + US[32 * nu * i:32 * nu * (i + 1)], gen_pts + UP[32 * nu * i:32 * nu * (i + 1)])
                assert rb[1][i] == 0 and rb[0][32 * i:32 * i + 32] == exp[1], (nb, nu, i)
        # an undecodable point of the caller: status 1, zero encoding, in both forms
        s, p = _rand_msm(oracle, b"bad", 17)
        p = p[:64] + bytes([p[64] | 1]) + p[65:]
        gs = b"".join(_scalar(b"bg%d" % j) for j in range(ngen))
        ra, rb = a.msm_batch_shared(n, m, 1, 17, gs, s, p), b.msm_batch_shared(n, m, 1, 17, gs, s, p)
        assert ra == rb and rb[1][0] == 1 and rb[0] == bytes(32)
    finally:
        a.close()
        b.close()

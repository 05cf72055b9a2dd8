"""One generator set, two shapes: BulletproofGens::new(64, 16) serves m = 16 AND m = 1 proofs (src/generators.rs:157-259).  The window
table is sized for the whole set; bpgpu_gens_add_shape / bpgpu_pool_gens_add_shape add a second table for the smaller shape and
re-balance both windows under one budget.  Whatever table a proof walks, verdicts and mega-check encodings equal the oracle's."""
import hashlib
import os
import threading

import pytest

pytestmark = pytest.mark.gpu

from test_gpu_bench_config import _tamper  # noqa: E402


@pytest.mark.parametrize("budget_gib,expect", [(8, (10, 15)), (160, (15, 19))])
def test_pool_two_shapes_one_budget_concurrent_vs_oracle(oracle, budget_gib, expect):
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    f2, f3 = wl.load_fixture("cfg2_n64_m1"), wl.load_fixture("cfg3_n64_m16")
    pool = bp.Pool((0,), 8, fixed_table_max_bytes=budget_gib << 30)
    pool.gens_create(64, 16)
    w_alone = pool.get_option("fixed_window_bits")
    pool.gens_add_shape(64, 1)
    w1, w2 = pool.get_option("fixed_window_bits"), pool.get_option("secondary_window_bits")
    assert (w1, w2) == expect and w1 <= w_alone
    assert pool.get_option("fixed_table_bytes") + pool.get_option("secondary_table_bytes") <= budget_gib << 30
    assert (pool.get_option("secondary_shape_n"), pool.get_option("secondary_shape_m")) == (64, 1)
    nb2, nb3 = 1500, 300
    p2, c2 = wl.tile_batch(f2, nb2, first=5)
    p2, c2, bad2 = _tamper(p2, c2, f2.proof_len, f2.m, nb2, 21)
    p3, c3 = wl.tile_batch(f3, nb3, first=3)
    p3, c3, bad3 = _tamper(p3, c3, f3.proof_len, f3.m, nb3, 22)
    r2, r3 = hashlib.shake_256(b"mix2").digest(64 * nb2), hashlib.shake_256(b"mix3").digest(64 * nb3)
    out = {}

    def run(key, fx, p, c, r):
        out[key] = pool.rangeproof_verify(fx.n, fx.m, p, fx.proof_len, c, fx.label, r, want_msm=True)

    th = [threading.Thread(target=run, args=("m1", f2, p2, c2, r2)), threading.Thread(target=run, args=("m16", f3, p3, c3, r3))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    gens = oracle.Gens(64, 16)
    for key, fx, p, c, r, bad in (("m1", f2, p2, c2, r2, bad2), ("m16", f3, p3, c3, r3, bad3)):
        _, ev, em = oracle.verify_batch(gens, p, c, fx.m, fx.n, fx.label, r, threads=os.cpu_count() or 1)
        v, msm = out[key]
        assert v == ev and sum(1 for x in ev if x) == len(bad), key
        for i in range(len(ev)):
            if ev[i] in (0, 1) and em[32 * i:32 * i + 32] != b"\xff" * 32:
                assert msm[32 * i:32 * i + 32] == em[32 * i:32 * i + 32], (key, i)
    # a shape the secondary table does not cover (m = 2) walks the primary one; same answers as a plain (64, 16) context
    gens8 = oracle.Gens(64, 16)
    pr, cm = oracle.prove(gens8, [5, 2 ** 40], bytes(64), 64, b"mixed", b"s")
    v = pool.rangeproof_verify(64, 2, pr * 3, len(pr), cm * 3, b"mixed", hashlib.shake_256(b"m2").digest(192))
    assert v == bytes(3)
    pool.close()


def test_context_add_shape_then_reload_drops_it(oracle):
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    f2 = wl.load_fixture("cfg2_n64_m1")
    ctx = bp.Context(0, fixed_table_max_bytes=2 << 30)
    ctx.gens_create(64, 4)
    ctx.gens_add_shape(64, 1)
    assert ctx.get_option("secondary_window_bits") > ctx.get_option("fixed_window_bits")
    p, c = wl.tile_batch(f2, 200)
    rng = hashlib.shake_256(b"ctx-mix").digest(64 * 200)
    v1, m1 = ctx.rangeproof_verify_batch(f2.n, f2.m, p, f2.proof_len, c, f2.label, rng, want_msm=True)
    ctx.gens_create(64, 4)                       # a new generator set: the secondary table is gone
    assert ctx.get_option("secondary_window_bits") == 0
    v2, m2 = ctx.rangeproof_verify_batch(f2.n, f2.m, p, f2.proof_len, c, f2.label, rng, want_msm=True)
    assert v1 == v2 == bytes(200) and m1 == m2
    with pytest.raises(bp.BpgpuError):
        ctx.gens_add_shape(64, 8)                # more parties than the set has
    ctx.close()

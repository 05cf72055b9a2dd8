"""The boundary function itself through the pool's combining queue: bpgpu_pool_msm_batch_shared / bpgpu_pool_msm_batch /
bpgpu_pool_ipp_verify (include/bpgpu.h) -- RistrettoPoint::optional_multiscalar_mul / vartime_multiscalar_mul as the crate calls it,
ONE multiscalar multiplication per call from whatever thread verifies (src/range_proof/mod.rs:421-445, src/r1cs/verifier.rs:459-491,
src/inner_product_proof.rs:308-319) and InnerProductProof::verify (ipp.rs:260-326).  Whatever chains the calls end up sharing,
every 32-byte encoding and status byte must equal the oracle's for that call's own inputs."""
import hashlib
import json
import os
import threading

import pytest

pytestmark = pytest.mark.gpu

from test_gpu_msm import _points, _rand_msm, _scalar  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pool16():
    import bulletproofs_amd as bp
    p = bp.Pool((0,), 4, fixed_window_bits=12)
    p.gens_create(16, 2)
    yield p
    p.close()


def _shared_case(oracle, G, H, B, Bb, n, m, nu, tag):
    ngen = 2 * n * m + 2
    cap = len(G) // 32 // 2   # (gens capacity 16, 2 parties: party-major)
    gen_pts = Bb + B + b"".join(G[32 * cap * j:32 * cap * j + 32 * n] for j in range(m)) + b"".join(H[32 * cap * j:32 * cap * j + 32 * n] for j in range(m))
    gs = b"".join(_scalar(b"%s-g%d" % (tag, i)) for i in range(ngen))
    us, up = _rand_msm(oracle, tag + b"-u", nu) if nu else (b"", b"")
    exp = oracle.msm(gs + us, gen_pts + up)
    return gs, us, up, exp


def test_single_msm_calls_from_many_threads_share_chains_and_match_the_oracle(oracle, pool16):
    """48 threads, each looping single-MSM bpgpu_pool_msm_batch_shared calls of two shapes (the r1cs verifier's call shape, scaled
    down): results == oracle, and the calls did share launch chains"""
    g = oracle.Gens(16, 2)
    G, H, B, Bb = g.export()
    cases = {}
    for k in range(12):
        shape = (16, 2, 9) if k % 2 == 0 else (8, 1, 33)
        cases[k] = (shape,) + _shared_case(oracle, G, H, B, Bb, shape[0], shape[1], shape[2], b"pm%d" % k)
    pool16.set_option("stat_reset", 1)
    errs = []

    def worker(t):
        try:
            for it in range(6):
                (n, m, nu), gs, us, up, exp = cases[(t + it) % 12]
                out, st = pool16.msm_batch_shared(n, m, 1, nu, gs, us, up)
                assert st[0] == exp[0] == 0 and out == exp[1], (t, it)
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(48)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs[:3]
    chains, items = pool16.get_option("stat_combined_chains"), pool16.get_option("stat_combined_proofs")
    assert items == 48 * 6 and chains < items, (chains, items)   # (some chains carried several callers' MSMs)


def test_batch_of_shared_msms_and_the_empty_unique_part(oracle, pool16):
    g = oracle.Gens(16, 2)
    G, H, B, Bb = g.export()
    nb = 70
    cs = [_shared_case(oracle, G, H, B, Bb, 16, 2, 9, b"pb%d" % b) for b in range(nb)]
    out, st = pool16.msm_batch_shared(16, 2, nb, 9, b"".join(c[0] for c in cs), b"".join(c[1] for c in cs), b"".join(c[2] for c in cs))
    for b in range(nb):
        assert st[b] == 0 and out[32 * b:32 * b + 32] == cs[b][3][1], b
    gs, us, up, exp = _shared_case(oracle, G, H, B, Bb, 8, 1, 0, b"nouniq")
    out, st = pool16.msm_batch_shared(8, 1, 1, 0, gs, b"", b"")
    assert st[0] == 0 and out == exp[1]
    # an undecodable point: status 1 (the reference's None), as on a context
    bad_up = b"\xff" * 32 + cs[0][2][32:]
    out, st = pool16.msm_batch_shared(16, 2, 1, 9, cs[0][0], cs[0][1], bad_up)
    assert st[0] == 1


def test_ragged_msm_batch_through_the_queue(oracle, pool16):
    """bpgpu_pool_msm_batch: stretches of equal term counts become classes; 0-term MSMs are the identity; many threads"""
    sizes = [0, 1, 1, 2, 31, 31, 31, 33, 64, 147, 147, 200, 1, 0, 700]
    S, P, exp = b"", b"", []
    for k, n in enumerate(sizes):
        s, p = _rand_msm(oracle, b"pr%d" % k, n)
        S += s
        P += p
        exp.append(oracle.msm(s, p))
    errs = []

    def worker(t):
        try:
            out, st = pool16.msm_batch(sizes, S, P)
            for k in range(len(sizes)):
                assert st[k] == 0 and out[32 * k:32 * k + 32] == exp[k][1], (t, k, sizes[k])
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs[:3]


def test_config5_shape_single_calls_against_committed_oracle_encodings(oracle):
    """BASELINE config 5's MSM (6179 terms) in the crate's call shape: 16 threads, one MSM per blocking call, through
    bpgpu_pool_msm_batch_shared -- every result == the oracle's committed encoding of that MSM (bench_data/cfg5_expected.json:
    tools/gen_cfg5_expected.py); the bucket path runs inside the shared chains"""
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    with open(os.path.join(ROOT, "bench_data", "cfg5_expected.json")) as f:
        exp = json.load(f)
    n, nu, nb = 2048, 2081, 64
    G2, H2, _, _ = oracle.Gens(n, 2).export()
    gs, us, up = wl.cfg5_inputs(G2, H2, nb)
    ng = 2 * n + 2
    pool = bp.Pool((0,), 4, fixed_window_bits=10)
    pool.gens_create(n, 1)
    errs = []

    def worker(t):
        try:
            for it in range(4):
                b = (t * 4 + it) % nb
                out, st = pool.msm_batch_shared(n, 1, 1, nu, gs[32 * ng * b:32 * ng * (b + 1)], us[32 * nu * b:32 * nu * (b + 1)], up[32 * nu * b:32 * nu * (b + 1)])
                assert st[0] == 0 and out.hex() == exp["msm%d" % b], (t, b)
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    [x.start() for x in th]
    [x.join() for x in th]
    chains, items = pool.get_option("stat_combined_chains"), pool.get_option("stat_combined_proofs")
    pool.close()
    assert not errs, errs[:3]
    assert items == 64 and chains < 64, (chains, items)


@pytest.mark.parametrize("n", [2, 32])
def test_inner_product_proofs_one_per_call_from_many_threads(oracle, n):
    """bpgpu_pool_ipp_verify == oracle.ipp_verify per proof (valid, tampered, wrong P, non-canonical, identity L_0), one proof per call
    from 24 threads; plus a batch call and the malformed-length path (reported by the ordinary entry point)"""
    import bulletproofs_amd as bp
    pool = bp.Pool((0,), 4, fixed_window_bits=8)
    pool.gens_create(8, 1)   # (the stand-alone inner-product check brings its own bases: the tables are not consulted)
    nb = 10
    insts = [oracle.ipp_test_instance(n, b"innerproducttest", b"pipp-%d-%d" % (n, j)) for j in range(nb)]
    pl = len(insts[0]["proof"])
    t = bytearray(insts[1]["proof"])
    t[-40] ^= 1
    insts[1] = dict(insts[1], proof=bytes(t))
    insts[2] = dict(insts[2], P=insts[2]["Q"])
    nc = bytearray(insts[3]["proof"])
    nc[-32:] = b"\xff" * 32
    insts[3] = dict(insts[3], proof=bytes(nc))
    li = bytearray(insts[4]["proof"])
    li[0:32] = bytes(32)
    insts[4] = dict(insts[4], proof=bytes(li))
    exp = [oracle.ipp_verify(n, i["proof"], b"innerproducttest", i["Gf"], i["Hf"], i["P"], i["Q"], i["G"], i["H"]) for i in insts]
    errs = []

    def worker(tix):
        try:
            for it in range(5):
                j = (tix + it) % nb
                i = insts[j]
                v, msm = pool.ipp_verify(n, i["proof"], pl, b"innerproducttest", i["Gf"], i["Hf"], i["P"], i["Q"], i["G"], i["H"], want_msm=True)
                assert v[0] == exp[j][0], (tix, j)
                if exp[j][0] != 2 and exp[j][1] != b"\xff" * 32 and j != 4:
                    assert msm == exp[j][1], (tix, j)
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(24)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs[:3]
    cat = lambda key: b"".join(i[key] for i in insts)
    v = pool.ipp_verify(n, cat("proof"), pl, b"innerproducttest", cat("Gf"), cat("Hf"), cat("P"), cat("Q"), cat("G"), cat("H"))
    assert list(v) == [e[0] for e in exp]
    i0 = insts[0]
    v = pool.ipp_verify(n, i0["proof"][:-1], pl - 1, b"x", i0["Gf"], i0["Hf"], i0["P"], i0["Q"], i0["G"], i0["H"])
    assert list(v) == [2]
    v = pool.ipp_verify(n, i0["proof"], pl, b"other label", i0["Gf"], i0["Hf"], i0["P"], i0["Q"], i0["G"], i0["H"])
    assert list(v) == [1]
    pool.close()


def test_device_resident_msm_batches_on_the_pool_lanes_with_tickets(oracle, pool16):
    """bpgpu_pool_msm_batch_shared_submit_dev: MSM batches whose inputs already sit in HBM go out one chain per batch on the pool's lanes
    (round-robin), ordered behind the producer's stream; a ticket per batch; results == oracle.  What bench.py's config-5 figure runs."""
    import torch
    dev = torch.device("cuda", 0)
    g = oracle.Gens(16, 2)
    G, H, B, Bb = g.export()
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    prod = torch.cuda.Stream(device=dev)
    batches = []
    for k in range(9):
        n, m, nu = ((16, 2, 9), (8, 1, 33), (16, 1, 0))[k % 3]
        nb = 1 + 5 * k
        cs = [_shared_case(oracle, G, H, B, Bb, n, m, nu, b"dv%d-%d" % (k, b)) for b in range(nb)]
        h = dict(n=n, m=m, nu=nu, nb=nb, exp=[c[3] for c in cs], gs=b"".join(c[0] for c in cs), us=b"".join(c[1] for c in cs), up=b"".join(c[2] for c in cs))
        with torch.cuda.stream(prod):   # the inputs are PRODUCED on another stream: the chain must wait for exactly that
            h["d_gs"] = torch.frombuffer(bytearray(h["gs"]), dtype=torch.uint8).pin_memory().to(dev, non_blocking=True)
            h["d_us"] = torch.frombuffer(bytearray(h["us"]), dtype=torch.uint8).pin_memory().to(dev, non_blocking=True) if nu else None
            h["d_up"] = torch.frombuffer(bytearray(h["up"]), dtype=torch.uint8).pin_memory().to(dev, non_blocking=True) if nu else None
        h["d_out"] = torch.full((nb, 32), 255, dtype=torch.uint8, device=dev)
        h["d_st"] = torch.full((nb,), 255, dtype=torch.uint8, device=dev)
        batches.append(h)
    torch.cuda.current_stream().synchronize()      # (the output buffers exist; the producer stream has NOT been waited for)
    tickets = []
    for h in batches:
        tickets.append(pool16.msm_shared_submit_dev(0, h["n"], h["m"], h["nb"], h["nu"], h["d_gs"].data_ptr(), h["d_us"].data_ptr() if h["nu"] else None,
                                                    h["d_up"].data_ptr() if h["nu"] else None, h["d_out"].data_ptr(), h["d_st"].data_ptr(),
                                                    producer_stream=prod.cuda_stream, want_ticket=True))
    for t in tickets:
        t.wait()
    for k, h in enumerate(batches):
        out, st = bytes(h["d_out"].cpu().numpy().reshape(-1)), bytes(h["d_st"].cpu().numpy())
        for b in range(h["nb"]):
            assert st[b] == h["exp"][b][0] == 0 and out[32 * b:32 * b + 32] == h["exp"][b][1], (k, b)
    # without tickets: bpgpu_pool_wait covers the lanes
    h = batches[4]
    h["d_out"].fill_(255)
    pool16.msm_shared_submit_dev(0, h["n"], h["m"], h["nb"], h["nu"], h["d_gs"].data_ptr(), h["d_us"].data_ptr(), h["d_up"].data_ptr(), h["d_out"].data_ptr(),
                                 h["d_st"].data_ptr())
    pool16.wait()
    assert bytes(h["d_out"].cpu().numpy().reshape(-1)) == b"".join(e[1] for e in h["exp"])

"""bpgpu_pool_* (include/bpgpu.h): the scheduler of the library -- ONE host-pointer call for any number of proofs, sliced
over the pool's lanes and sharded over its devices, and asynchronous device-pointer batches coalesced into wide launch
chains.  The call shape replaced: a loop over RangeProof::verify_multiple (src/range_proof/mod.rs:457-470).  Every verdict
byte (and mega-check encoding) must equal the oracle's, whatever the slicing / coalescing."""
import hashlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

from test_gpu_bench_config import _tamper  # noqa: E402


def _oracle_all(oracle, gens, fx, proofs, coms, rng):
    _, ev, em = oracle.verify_batch(gens, proofs, coms, fx.m, fx.n, fx.label, rng, threads=os.cpu_count() or 1)
    return ev, em


def _msm_equal(ev, em, msm):
    for i in range(len(ev)):   # the oracle leaves 0xff.. for proofs it rejects before the MSM
        if ev[i] in (0, 1) and em[32 * i:32 * i + 32] != b"\xff" * 32:
            assert msm[32 * i:32 * i + 32] == em[32 * i:32 * i + 32], i


@pytest.fixture(scope="module")
def cfg2():
    from bulletproofs_amd import workload as wl
    return wl.load_fixture("cfg2_n64_m1")


def test_pool_one_call_65536_tampered_proofs_vs_oracle(oracle, cfg2):
    """ONE call, ONE host thread, host pointers, 65 536 proofs (~5 % tampered seven ways): verdicts == oracle."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = cfg2
    nb = 65536
    proofs, coms = wl.tile_batch(fx, nb)
    proofs, coms, bad = _tamper(proofs, coms, fx.proof_len, fx.m, nb, 77)
    rng = hashlib.shake_256(b"pool-65536").digest(64 * nb)
    pool = bp.Pool((0,), 16)
    pool.gens_create(64, 1)
    verdict = pool.rangeproof_verify(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng)
    ev, _ = _oracle_all(oracle, oracle.Gens(64, 1), fx, proofs, coms, rng)
    assert verdict == ev
    assert sum(1 for v in ev if v) == len(bad)
    # library-drawn randomness (rng64 = NULL, the thread_rng() of verify_multiple): same verdicts
    assert pool.rangeproof_verify(fx.n, fx.m, proofs[:5000 * fx.proof_len], fx.proof_len, coms[:5000 * 32], fx.label) == ev[:5000]
    pool.close()


@pytest.mark.parametrize("devices,lanes,slice_proofs", [((0,), 4, 0), ((0, 0), 3, 700), ((0, 0, 0), 2, 64)])
def test_pool_slicing_and_sharding_bit_exact(oracle, cfg2, devices, lanes, slice_proofs):
    """Ragged sizes, explicit slice widths, several shards (two / three pool devices on the one GPU): verdicts AND encodings
    == oracle; an empty call and a one-proof call work."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = cfg2
    pool = bp.Pool(devices, lanes, fixed_window_bits=16)
    if slice_proofs:
        pool.set_option("slice_proofs", slice_proofs)
    pool.gens_create(64, 1)
    assert pool.n_devices == len(devices) and pool.n_lanes == lanes
    gens = oracle.Gens(64, 1)
    for nb, seed in ((0, 1), (1, 2), (3001, 3)):
        proofs, coms = wl.tile_batch(fx, nb, first=100 * seed)
        if nb > 1:
            proofs, coms, _ = _tamper(proofs, coms, fx.proof_len, fx.m, nb, seed)
        rng = hashlib.shake_256(b"pool-%d" % seed).digest(64 * nb)
        verdict, msm = pool.rangeproof_verify(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng, want_msm=True)
        if nb == 0:
            assert verdict == b"" and msm == b""
            continue
        ev, em = _oracle_all(oracle, gens, fx, proofs, coms, rng)
        assert verdict == ev
        _msm_equal(ev, em, msm)
    pool.close()


def test_pool_coalesced_device_batches_vs_oracle(oracle, cfg2):
    """submit_dev: batches of different sizes (1, 300, 1024, 2048, 77), some with their own rng bytes and some without, some
    wanting encodings: coalesced into chains of ~1500 proofs (items split across chains), plus a malformed-length batch and
    a parameter-error batch in the same flush; every batch's own verdict buffer == oracle."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = cfg2
    dev = torch.device("cuda", 0)
    pool = bp.Pool((0,), 4, fixed_window_bits=16)
    pool.set_option("coalesce_proofs", 1500)
    pool.set_option("auto_flush_items", 1000)
    pool.gens_create(64, 1)
    gens = oracle.Gens(64, 1)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    sizes = [1, 300, 1024, 2048, 77, 513]
    items, first = [], 0
    for i, nb in enumerate(sizes):
        proofs, coms = wl.tile_batch(fx, nb, first=first)
        first += nb
        if nb > 1:
            proofs, coms, _ = _tamper(proofs, coms, fx.proof_len, fx.m, nb, 10 + i, frac=0.1)
        rng = hashlib.shake_256(b"co-%d" % i).digest(64 * nb)
        own_rng, want_msm = (i % 2 == 0), (i % 3 != 2)
        d = dict(nb=nb, proofs=proofs, coms=coms, rng=rng, d_p=to_dev(proofs), d_c=to_dev(coms), d_r=to_dev(rng) if own_rng else None,
                 d_v=torch.full((nb,), 255, dtype=torch.uint8, device=dev),
                 d_m=torch.full((nb, 32), 255, dtype=torch.uint8, device=dev) if want_msm else None)
        items.append(d)
    # a batch whose proofs have a malformed length (every proof FormatError) and one checked against n = 32 (parameter error)
    bad_len = fx.proof_len - 32
    d_bl = to_dev(items[2]["proofs"][:bad_len * 5])
    d_blv = torch.full((5,), 255, dtype=torch.uint8, device=dev)
    d_n32v = torch.full((300,), 255, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for rounds in range(2):
        for i, d in enumerate(items):
            pool.submit_dev(0, fx.n, fx.m, d["nb"], d["d_p"].data_ptr(), fx.proof_len, d["d_c"].data_ptr(), fx.label,
                            d["d_r"].data_ptr() if d["d_r"] is not None else None, d["d_v"].data_ptr(), d["d_m"].data_ptr() if d["d_m"] is not None else None)
            if i == 2:
                pool.submit_dev(0, fx.n, fx.m, 5, d_bl.data_ptr(), bad_len, items[2]["d_c"].data_ptr(), fx.label, None, d_blv.data_ptr(), None)
            if i == 3:
                pool.submit_dev(0, 32, fx.m, 300, items[1]["d_p"].data_ptr(), fx.proof_len, items[1]["d_c"].data_ptr(), fx.label, None, d_n32v.data_ptr(), None)
        pool.wait()
        for d in items:
            ev, em = _oracle_all(oracle, gens, fx, d["proofs"], d["coms"], d["rng"])
            assert bytes(d["d_v"].cpu().numpy()) == ev
            if d["d_m"] is not None and d["d_r"] is not None:     # encodings depend on the rng bytes: comparable when the item brought its own
                _msm_equal(ev, em, bytes(d["d_m"].cpu().numpy().reshape(-1)))
            d["d_v"].fill_(255)
        assert bytes(d_blv.cpu().numpy()) == bytes([2] * 5)                      # FormatError, as RangeProof::from_bytes
        # n = 32 against 64-bit proofs: n * m != 2^k -> VerificationError for well-formed proofs (ipp.rs:209), Format for the others
        ev1, _ = _oracle_all(oracle, gens, fx, items[1]["proofs"], items[1]["coms"], items[1]["rng"])
        assert bytes(d_n32v.cpu().numpy()) == bytes(2 if v == 2 else 1 for v in ev1)
        torch.cuda.synchronize()
    pool.close()


def test_pool_burst_of_20_batches_coalesced_equals_per_batch_path(oracle, cfg2):
    """The driver's burst (20 batches of 1024 submitted back to back, then one flush -- two chains of 10 240 at the default
    W = 20): every batch's verdicts are what bpgpu_rangeproof_verify_batch_dev gives for it alone, and verdicts AND 32-byte
    mega-check encodings equal the oracle's on two of them."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = cfg2
    dev = torch.device("cuda", 0)
    pool = bp.Pool((0,), 32)
    pool.gens_create(64, 1)
    ctx = bp.Context(0)
    ctx.gens_create(64, 1)
    L = bp.lib()
    K, nb = 20, 1024
    proofs, coms = wl.tile_batch(fx, K * nb, first=123)
    proofs, coms, bad = _tamper(proofs, coms, fx.proof_len, fx.m, K * nb, 99, frac=0.02)
    rng = hashlib.shake_256(b"burst").digest(64 * K * nb)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_p, d_c, d_r = to_dev(proofs), to_dev(coms), to_dev(rng)
    d_v = torch.full((K, nb), 255, dtype=torch.uint8, device=dev)
    d_ref = torch.full((K, nb), 255, dtype=torch.uint8, device=dev)
    d_m = torch.zeros((K, nb, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    pool.set_option("stat_reset", 1)
    for k in range(K):
        pool.submit_dev(0, fx.n, fx.m, nb, d_p.data_ptr() + k * nb * fx.proof_len, fx.proof_len, d_c.data_ptr() + k * nb * 32, fx.label,
                        d_r.data_ptr() + k * nb * 64, d_v[k].data_ptr(), d_m[k].data_ptr())
    pool.wait()
    assert pool.get_option("stat_chains") == 2 and pool.get_option("stat_chain_proofs") == K * nb   # the benched form: two chains of 10 240
    for k in range(K):
        rc = L.bpgpu_rangeproof_verify_batch_dev(ctx.h, fx.n, fx.m, nb, d_p.data_ptr() + k * nb * fx.proof_len, fx.proof_len, d_c.data_ptr() + k * nb * 32,
                                                 fx.label, len(fx.label), d_r.data_ptr() + k * nb * 64, d_ref[k].data_ptr(), None, None)
        assert rc == 0
    ctx.synchronize()
    assert bool((d_v == d_ref).all().item())
    assert int((d_v != 0).sum().item()) == len(bad)
    gens = oracle.Gens(64, 1)
    for k in (0, 13):
        ev, em = _oracle_all(oracle, gens, fx, proofs[k * nb * fx.proof_len:(k + 1) * nb * fx.proof_len], coms[k * nb * 32:(k + 1) * nb * 32],
                             rng[k * nb * 64:(k + 1) * nb * 64])
        assert bytes(d_v[k].cpu().numpy()) == ev
        _msm_equal(ev, em, bytes(d_m[k].cpu().numpy().reshape(-1)))
    ctx.close()
    pool.close()


@pytest.mark.parametrize("ndev", [1, 2])
def test_plain_c_client_of_the_pool(tmp_path, ndev):
    """tests/cpp/pool_client.c, compiled with gcc against include/bpgpu.h only: one bpgpu_pool_rangeproof_verify call for 6000
    proofs of the cfg2 fixture on one / two shards, library-drawn randomness; the rejected indices are exactly the planted ones."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "pool_client"
    libdir = os.path.join(root, "bulletproofs_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "pool_client.c"),
                           "-o", str(exe), "-L", libdir, "-lbpgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe), os.path.join(root, "bench_data", "cfg2_n64_m1.bin"), str(ndev), "6000", "487"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    planted = [i for i in range(6000) if i % 487 == 486]
    assert out.stdout.strip() == "rejected %d of 6000:" % len(planted) + "".join(" %d=1" % i for i in planted)


def test_pool_shared_by_threads_host_and_device_paths_mixed(oracle, cfg2):
    """One pool used from four host threads at once -- host-pointer calls of different sizes and device-pointer submissions with
    their own flush / wait -- while the pool's own workers run: no deadlock, every verdict == oracle (the pool serialises its
    bookkeeping internally; lanes are shared by both paths)."""
    import threading
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = cfg2
    dev = torch.device("cuda", 0)
    pool = bp.Pool((0, 0), 6, fixed_window_bits=16)
    pool.gens_create(64, 1)
    gens = oracle.Gens(64, 1)
    sizes = (1500, 700, 2600, 333)
    cases = []
    for i, nb in enumerate(sizes):
        proofs, coms = wl.tile_batch(fx, nb, first=977 * i)
        proofs, coms, _ = _tamper(proofs, coms, fx.proof_len, fx.m, nb, 40 + i, frac=0.03)
        rng = hashlib.shake_256(b"mt-%d" % i).digest(64 * nb)
        ev, _ = _oracle_all(oracle, gens, fx, proofs, coms, rng)
        cases.append((nb, proofs, coms, rng, ev))
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    dv = [(to_dev(c[1]), to_dev(c[2]), to_dev(c[3]), torch.full((c[0],), 255, dtype=torch.uint8, device=dev)) for c in cases]
    torch.cuda.synchronize()
    errors = []

    def host_worker(i):
        try:
            nb, proofs, coms, rng, ev = cases[i]
            for _ in range(3):
                if pool.rangeproof_verify(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng) != ev:
                    errors.append("host %d" % i)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    def dev_worker():
        try:
            for _ in range(3):
                for i, (nb, *_rest) in enumerate(cases):
                    pool.submit_dev(i % 2, fx.n, fx.m, nb, dv[i][0].data_ptr(), fx.proof_len, dv[i][1].data_ptr(), fx.label, dv[i][2].data_ptr(), dv[i][3].data_ptr())
                pool.wait()
                for i, c in enumerate(cases):
                    if bytes(dv[i][3].cpu().numpy()) != c[4]:
                        errors.append("dev %d" % i)
                    dv[i][3].fill_(255)
                torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=host_worker, args=(i,)) for i in range(3)] + [threading.Thread(target=dev_worker)]
    [t.start() for t in ths]
    [t.join(timeout=300) for t in ths]
    assert not any(t.is_alive() for t in ths), "deadlock"
    assert errors == []
    pool.close()


def test_pool_many_small_items_in_one_chain_and_flush_by_proofs(oracle, cfg2):
    """40 submitted batches of 53 proofs each: one flush packs them into chains of up to 40 segments (more than the 16 that travel in
    the kernels' argument blocks: the segment table goes through device memory); then the same with auto_flush_proofs = 1000 (a chain
    leaves as soon as 1000 proofs wait, while the caller is still submitting).  Every batch's verdicts == oracle, chain counts as designed."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = cfg2
    dev = torch.device("cuda", 0)
    pool = bp.Pool((0,), 8, fixed_window_bits=16)
    pool.set_option("auto_flush_items", 1000)
    pool.gens_create(64, 1)
    gens = oracle.Gens(64, 1)
    K, nb = 40, 53
    proofs, coms = wl.tile_batch(fx, K * nb, first=4000)
    proofs, coms, bad = _tamper(proofs, coms, fx.proof_len, fx.m, K * nb, 7, frac=0.04)
    rng = hashlib.shake_256(b"small-items").digest(64 * K * nb)
    ev, _ = _oracle_all(oracle, gens, fx, proofs, coms, rng)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_p, d_c, d_r = to_dev(proofs), to_dev(coms), to_dev(rng)
    d_v = torch.full((K, nb), 255, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for afp, chains in ((0, 1), (1000, 3)):     # 2120 proofs: one chain (coalesce_proofs 5120) / 1007 + 1007 by the proof count + the rest at the flush
        pool.set_option("auto_flush_proofs", afp)
        pool.set_option("stat_reset", 1)
        for k in range(K):
            pool.submit_dev(0, fx.n, fx.m, nb, d_p.data_ptr() + k * nb * fx.proof_len, fx.proof_len, d_c.data_ptr() + k * nb * 32, fx.label,
                            d_r.data_ptr() + k * nb * 64, d_v[k].data_ptr())
        pool.wait()
        assert bytes(d_v.cpu().numpy().reshape(-1)) == ev
        assert pool.get_option("stat_chains") == chains and pool.get_option("stat_chain_proofs") == K * nb
        d_v.fill_(255)
        torch.cuda.synchronize()
    assert sum(1 for v in ev if v) == len(bad)
    pool.close()


def test_submit_dev_ex_producer_stream_and_tickets_no_device_wide_sync(oracle, cfg2):
    """The completion contract of bpgpu_pool_rangeproof_submit_dev_ex: inputs produced on the caller's stream (behind ~ms of other work,
    so a chain that did not wait would read garbage), one ticket per batch, a consumer stream that waits for THAT batch on the device
    and copies its verdicts out -- batch k is consumed while batches k+1.. are queued or running; the host only ever waits on the
    consumer stream.  Verdicts == oracle."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = cfg2
    dev = torch.device("cuda", 0)
    pool = bp.Pool((0,), 8, fixed_window_bits=16, auto_flush_items=3)
    pool.gens_create(64, 1)
    K, nb = 7, 700
    proofs, coms = wl.tile_batch(fx, K * nb, first=31)
    proofs, coms, bad = _tamper(proofs, coms, fx.proof_len, fx.m, K * nb, 4242, frac=0.03)
    rng = hashlib.shake_256(b"pc").digest(64 * K * nb)
    pin = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).pin_memory()
    h_p, h_c, h_r = pin(proofs), pin(coms), pin(rng)
    d_p = torch.full((K * nb * fx.proof_len,), 0xA5, dtype=torch.uint8, device=dev)     # garbage until the producer's copy lands
    d_c = torch.full((K * nb * 32,), 0xA5, dtype=torch.uint8, device=dev)
    d_r = torch.zeros((K * nb * 64,), dtype=torch.uint8, device=dev)
    d_v = torch.full((K, nb), 255, dtype=torch.uint8, device=dev)
    h_v = torch.full((K, nb), 254, dtype=torch.uint8).pin_memory()
    busy = torch.randn((4096, 4096), device=dev)
    producer, consumer = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    tickets = []
    for k in range(K):
        with torch.cuda.stream(producer):
            for _ in range(3):
                busy = busy @ busy * 1e-4                      # ~ms of unrelated work ahead of the inputs on the producer's stream
            sl = slice(k * nb * fx.proof_len, (k + 1) * nb * fx.proof_len)
            d_p[sl].copy_(h_p[sl], non_blocking=True)
            d_c[k * nb * 32:(k + 1) * nb * 32].copy_(h_c[k * nb * 32:(k + 1) * nb * 32], non_blocking=True)
            d_r[k * nb * 64:(k + 1) * nb * 64].copy_(h_r[k * nb * 64:(k + 1) * nb * 64], non_blocking=True)
        tickets.append(pool.submit_dev_ex(0, fx.n, fx.m, nb, d_p.data_ptr() + k * nb * fx.proof_len, fx.proof_len, d_c.data_ptr() + k * nb * 32, fx.label,
                                          d_r.data_ptr() + k * nb * 64, d_v[k].data_ptr(), producer_stream=producer.cuda_stream))
        if k >= 2:   # consume batch k-2 while k-1 and k are queued / running
            j = k - 2
            tickets[j].stream_wait(consumer.cuda_stream)
            with torch.cuda.stream(consumer):
                h_v[j].copy_(d_v[j], non_blocking=True)
    for j in (K - 2, K - 1):
        tickets[j].stream_wait(consumer.cuda_stream)
        with torch.cuda.stream(consumer):
            h_v[j].copy_(d_v[j], non_blocking=True)
    consumer.synchronize()                                       # the only host-side wait: no hipDeviceSynchronize, no pool.wait()
    gens = oracle.Gens(64, 1)
    ev, _ = _oracle_all(oracle, gens, fx, proofs, coms, rng)
    assert bytes(h_v.numpy().reshape(-1)) == ev
    assert sum(1 for v in ev if v) == len(bad)
    assert all(t.done() for t in tickets)
    for t in tickets:
        t.wait()                                                 # frees the tickets (everything is complete already)
    pool.close()


def test_pool_create_asks_the_device_when_the_library_set_the_queue_variable_itself():
    """GPU_MAX_HW_QUEUES is read by the ROCm runtime at the process's first HIP call.  Unset in the environment, libbpgpu sets it when it
    is loaded -- which helps only if nothing initialised HIP earlier.  (a) library first: the pool gets its queues (the probe sees sixteen
    spinning kernels overlap) and works; (b) HIP first (here: a hipMalloc through ctypes on libamdhip64 before libbpgpu is loaded): the
    variable still reads 16 afterwards, but the runtime runs on its default 4 queues -- bpgpu_pool_create must say so
    (BPGPU_ERR_HW_QUEUES = -6), not build 8 lanes that silently serialise."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    prog = ("import ctypes as C, os, sys\n"
            "hip_first = sys.argv[1] == '1'\n"
            "if hip_first:\n"
            "    h = C.CDLL('libamdhip64.so'); p = C.c_void_p(); assert h.hipMalloc(C.byref(p), 1024) == 0\n"
            "L = C.CDLL(os.path.join(%r, 'bulletproofs_amd', 'csrc', 'libbpgpu.so'))\n"
            "pool = C.c_void_p(); d = (C.c_int * 1)(0)\n"
            "rc = L.bpgpu_pool_create(d, 1, 8, C.byref(pool))\n"
            "g = C.CDLL(None).getenv; g.restype = C.c_char_p\n"
            "print('rc', rc, (g(b'GPU_MAX_HW_QUEUES') or b'None').decode())\n" % root)
    a = subprocess.run([sys.executable, "-c", prog, "0"], env=env, capture_output=True, text=True, timeout=300)
    assert a.stdout.strip().endswith("rc 0 16"), (a.stdout, a.stderr[-400:])
    b = subprocess.run([sys.executable, "-c", prog, "1"], env=env, capture_output=True, text=True, timeout=300)
    assert b.stdout.strip().endswith("rc -6 16"), (b.stdout, b.stderr[-400:])
    assert "HIP had been initialised before" in b.stderr


def test_device_resident_verdicts_gathered_on_one_device(oracle, cfg2):
    """Three shards (the one GPU three times), device-resident batches submitted per shard, verdicts gathered into ONE device buffer on
    shard 1 by bpgpu_pool_gather_dev -- no host copy of a verdict, no pool-wide wait: the gather stream is ordered behind the shards' lanes
    on the device.  == oracle."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = cfg2
    dev = torch.device("cuda", 0)
    pool = bp.Pool((0, 0, 0), 4, fixed_window_bits=16)
    pool.gens_create(64, 1)
    sizes = [700, 0, 1300]                       # a shard with nothing to do is skipped
    total = sum(sizes)
    proofs, coms = wl.tile_batch(fx, total, first=77)
    proofs, coms, bad = _tamper(proofs, coms, fx.proof_len, fx.m, total, 8)
    rng = hashlib.shake_256(b"gather").digest(64 * total)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_p, d_c, d_r = to_dev(proofs), to_dev(coms), to_dev(rng)
    parts = [torch.full((max(s, 1),), 255, dtype=torch.uint8, device=dev) for s in sizes]
    d_all = torch.full((total,), 254, dtype=torch.uint8, device=dev)
    gs = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    off = 0
    for di, nb in enumerate(sizes):
        if nb:
            pool.submit_dev(di, fx.n, fx.m, nb, d_p.data_ptr() + off * fx.proof_len, fx.proof_len, d_c.data_ptr() + off * 32, fx.label, d_r.data_ptr() + off * 64,
                            parts[di].data_ptr())
        off += nb
    pool.flush()
    pool.gather_dev(1, [t.data_ptr() for t in parts], sizes, d_all.data_ptr(), gs.cuda_stream)
    gs.synchronize()
    _, ev, _ = oracle.verify_batch(oracle.Gens(64, 1), proofs, coms, fx.m, fx.n, fx.label, rng, threads=os.cpu_count() or 1)
    assert bytes(d_all.cpu().numpy()) == ev and sum(1 for v in ev if v) == len(bad)
    pool.close()


def test_gather_between_distinct_devices_over_the_peer_links(oracle, cfg2):
    """The same gather with the shards on DIFFERENT GPUs (hipMemcpyPeerAsync over xGMI on an MI355X node): skipped on a one-GPU box, so that
    the day a node runs this suite the multi-device path has a parity test of its own."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = cfg2
    nd = min(torch.cuda.device_count(), 4)
    pool = bp.Pool(tuple(range(nd)), 4, fixed_window_bits=16)
    pool.gens_create(64, 1)
    sizes = [600 + 100 * d for d in range(nd)]
    total = sum(sizes)
    proofs, coms = wl.tile_batch(fx, total, first=5)
    proofs, coms, bad = _tamper(proofs, coms, fx.proof_len, fx.m, total, 9)
    rng = hashlib.shake_256(b"gather-peer").digest(64 * total)
    parts, keep, off = [], [], 0
    for d, nb in enumerate(sizes):
        dev = torch.device("cuda", d)
        to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        d_p, d_c, d_r = to_dev(proofs[off * fx.proof_len:(off + nb) * fx.proof_len]), to_dev(coms[off * 32:(off + nb) * 32]), to_dev(rng[off * 64:(off + nb) * 64])
        d_v = torch.full((nb,), 255, dtype=torch.uint8, device=dev)
        keep.append((d_p, d_c, d_r))
        parts.append(d_v)
        off += nb
    for d in range(nd):
        torch.cuda.synchronize(d)
    for d, nb in enumerate(sizes):
        d_p, d_c, d_r = keep[d]
        pool.submit_dev(d, fx.n, fx.m, nb, d_p.data_ptr(), fx.proof_len, d_c.data_ptr(), fx.label, d_r.data_ptr(), parts[d].data_ptr())
    pool.flush()
    root = nd - 1
    d_all = torch.full((total,), 254, dtype=torch.uint8, device=torch.device("cuda", root))
    gs = torch.cuda.Stream(device=torch.device("cuda", root))
    pool.gather_dev(root, [t.data_ptr() for t in parts], sizes, d_all.data_ptr(), gs.cuda_stream)
    gs.synchronize()
    _, ev, _ = oracle.verify_batch(oracle.Gens(64, 1), proofs, coms, fx.m, fx.n, fx.label, rng, threads=os.cpu_count() or 1)
    assert bytes(d_all.cpu().numpy()) == ev and sum(1 for v in ev if v) == len(bad)
    pool.close()


def test_every_documented_pool_option_flips_on_a_live_pool(oracle, cfg2):
    """include/bpgpu.h's option table, key by key: set a non-default value, read it back, verify a batch with it in force (verdicts ==
    oracle), set the default back.  Keys the table does not list are refused unless a lane context knows them; the keys removed in round 6
    are refused."""
    import bulletproofs_amd as bp
    sys.path.insert(0, os.path.dirname(__file__))
    from test_abi_and_host import _pool_option_table
    fx = cfg2
    nb = 96
    proofs = bytearray(fx.proofs[:nb * fx.proof_len])
    proofs[7 * fx.proof_len + 200] ^= 4
    proofs = bytes(proofs)
    coms = fx.commitments[:nb * 32 * fx.m]
    rng = hashlib.shake_256(b"opt-rng").digest(64 * nb)
    gens = oracle.Gens(64, 1)
    _, ev, _ = oracle.verify_batch(gens, proofs, coms, fx.m, fx.n, fx.label, rng, threads=4)
    pool = bp.Pool((0,), 4, fixed_window_bits=8)
    pool.gens_create(64, 1)
    other = {"coalesce_proofs": 2048, "max_chain_proofs": 8192, "pair_limit_proofs": 4096, "latency_proofs": 64, "auto_flush_items": 3, "auto_flush_proofs": 512,
             "slice_proofs": 32, "host_workers": 2, "host_path_combining": 0, "rlc_isolate": 1, "combine_wait_us": 250, "combine_quiet_us": 7,
             "combine_max_age_us": 900, "combine_inflight": 2, "combine_busy_chains": 1, "combine_max_open": 2, "combine_mapped_out": 0,
             "combine_msm_bytes": 1 << 20, "combine_trace": 64}
    table = _pool_option_table()
    assert {k for k, _ in table} == set(other) | {"stat_reset"}
    for key, _ in table:
        if key == "stat_reset":
            pool.set_option(key, 1)
            assert pool.get_option("stat_chains") == 0
            continue
        before = pool.get_option(key)
        assert before != other[key], key
        pool.set_option(key, other[key])
        assert pool.get_option(key) == other[key], key
        assert pool.rangeproof_verify(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng) == ev, key
        if key != "host_workers":          # (fixed once the worker threads of the slicing path exist)
            pool.set_option(key, before)
            assert pool.get_option(key) == before, key
    for gone in ("plan_by_work", "stagger_chains", "combine_policy", "combine_mapped_in", "split_stage1", "fork_early", "no_such_key"):
        with pytest.raises(Exception):
            pool.set_option(gone, 1)
    pool.close()

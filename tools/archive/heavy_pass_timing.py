"""Timing of the crowded-bucket (heavy) pass: the batch-combined check with short (128-bit) caller weights, and MSMs with equal scalars."""
import ctypes as C, hashlib, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bulletproofs_amd as bp
from bulletproofs_amd import workload as wl
L = bp.lib()
fx = wl.load_fixture("cfg2_n64_m1")
dev = torch.device("cuda", 0)
to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
for nb in (2048, 4096, 8192):
    proofs, coms = wl.tile_batch(fx, nb)
    d_p, d_c = to_dev(proofs), to_dev(coms)
    d_rng = to_dev(hashlib.shake_256(b"r").digest(64 * nb))
    w16 = hashlib.shake_256(b"w").digest(16 * nb)
    d_short = to_dev(b"".join(w16[16 * i:16 * i + 16] + bytes(48) for i in range(nb)))
    d_full = to_dev(hashlib.shake_256(b"wf").digest(64 * nb))
    d_v = torch.zeros(nb, dtype=torch.uint8, device=dev)
    c = bp.Context(0); c.gens_create(64, 1)
    st = torch.cuda.Stream(device=dev)
    for name, w in (("full", d_full), ("short128", d_short)):
        def call():
            rc = L.bpgpu_rangeproof_verify_rlc_dev(c.h, fx.n, fx.m, nb, d_p.data_ptr(), fx.proof_len, d_c.data_ptr(), fx.label, len(fx.label), d_rng.data_ptr(), w.data_ptr(), d_v.data_ptr(), None, st.cuda_stream)
            assert rc == 0, L.bpgpu_last_error(c.h)
        for _ in range(3): call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        assert int(d_v.sum()) == 0
        print("rlc nb=%d weights=%s: %.3f ms per combination (one stream)" % (nb, name, dt * 1e3))
    c.close()
# MSM with equal scalars
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyoracle as O
def pts(n):
    out = C.create_string_buffer(32); base = []
    for i in range(64):
        O.lib().oracle_from_uniform_bytes(hashlib.shake_256(b"p%d" % i).digest(64), out); base.append(out.raw)
    return b"".join(base[i % 64] for i in range(n))
Lq = 2**252 + 27742317777372353535851937790883648493
for n in (4000, 20000, 65536):
    P = pts(n)
    c = bp.Context(0); c.set_option("bucket_min_terms", 1)
    for name, S in (("uniform", b"".join((int.from_bytes(hashlib.shake_256(b"s%d" % i).digest(40), "little") % Lq).to_bytes(32, "little") for i in range(n))),
                    ("equal", (int.from_bytes(hashlib.shake_256(b"eq").digest(40), "little") % Lq).to_bytes(32, "little") * n)):
        c.msm_batch([n], S, P)
        t0 = time.perf_counter()
        for _ in range(5): out, st = c.msm_batch([n], S, P)
        dt = (time.perf_counter() - t0) / 5
        print("msm n=%d scalars=%s: %.3f ms per host call" % (n, name, dt * 1e3))
    c.close()

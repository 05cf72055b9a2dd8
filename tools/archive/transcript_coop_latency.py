"""Latency of one blocking host-pointer call of 1 / 8 / 64 / 256 proofs: default launch 1 against option transcript_coop."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bulletproofs_amd as bp
from bulletproofs_amd import workload as wl
fx = wl.load_fixture("cfg2_n64_m1")
ctxs = {}
for name, opt in (("default", 0), ("coop", 1)):
    c = bp.Context(0); c.set_option("transcript_coop", opt); c.gens_create(64, 1); ctxs[name] = c
for nb in (1, 8, 64, 256):
    proofs, coms = wl.tile_batch(fx, nb)
    rng = hashlib.shake_256(b"r").digest(64 * nb)
    for rep in range(2):
        for name, c in ctxs.items():
            for _ in range(20): v = c.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng)
            assert v == bytes(nb)
            ts = []
            for _ in range(200):
                t0 = time.perf_counter(); c.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng); ts.append(time.perf_counter() - t0)
            ts.sort()
            print("nb=%d %s: p50 %.3f ms  p10 %.3f ms" % (nb, name, ts[100] * 1e3, ts[20] * 1e3), flush=True)

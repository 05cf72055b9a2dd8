#!/usr/bin/env python3
"""Dev tool: thread scaling of the oracle's batch verifier on this host."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import pyoracle as O
from bulletproofs_amd.workload import load_fixture, tile_batch
fx = load_fixture("cfg2_n64_m1"); g = O.Gens(64, 1)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    nb = max(64, th * 24)
    proofs, coms = tile_batch(fx, nb); rng = hashlib.shake_256(b"x").digest(64 * nb)
    secs, v, _ = O.verify_batch(g, proofs, coms, 1, 64, fx.label, rng, threads=th)
    print("threads %3d: %6d proofs %.3fs -> %.0f verif/s (%.0f per thread)" % (th, nb, secs, nb / secs, nb / secs / th))

#!/usr/bin/env python3
"""Dev tool: PCIe-inclusive rate of the HOST-pointer entry point bpgpu_rangeproof_verify_batch
(H2D of proofs/commitments/rng, kernels, D2H of verdicts, allocation per call).  Not the bench metric."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bulletproofs_amd as bp
from bulletproofs_amd.workload import load_fixture, tile_batch
for cfg, batch in (("cfg2_n64_m1", 1024), ("cfg2_n64_m1", 16384), ("cfg3_n64_m16", 256)):
    fx = load_fixture(cfg)
    ctx = bp.Context(0); ctx.gens_create(fx.n, fx.m)
    proofs, coms = tile_batch(fx, batch); rng = hashlib.shake_256(b"h").digest(64 * batch)
    for _ in range(3): v = ctx.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps): v = ctx.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng)
    dt = (time.perf_counter() - t0) / reps
    assert v == bytes(batch)
    print("%s batch %5d: host-pointer API %.3f ms/call -> %.0f verifications/s (PCIe-inclusive)" % (cfg, batch, dt * 1e3, batch / dt))
    ctx.close()

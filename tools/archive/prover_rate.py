#!/usr/bin/env python3
"""Dev tool (GPU box): throughput of bpgpu_rangeproof_prove_batch (host pointers, one context) at a few shapes, next to the
oracle's prover on the host cores."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import bulletproofs_amd as bp
import pyoracle as O
for n, m, nb in ((64, 1, 1024), (64, 1, 4096), (64, 16, 256)):
    ctx = bp.Context(0)
    ctx.gens_create(n, m)
    vals = [int.from_bytes(hashlib.shake_256(b"v%d" % i).digest(8), "little") % (1 << n) for i in range(nb * m)]
    bl = hashlib.shake_256(b"bl").digest(32 * nb * m)
    per = 64 * (m * (2 * n + 2) + 2 * m)
    rng = hashlib.shake_256(b"rng").digest(per * nb)
    ctx.rangeproof_prove_batch(n, m, vals, bl, label=b"rate", rng=rng)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        proofs, coms = ctx.rangeproof_prove_batch(n, m, vals, bl, label=b"rate", rng=rng)
    dt = (time.perf_counter() - t0) / reps
    pl = 32 * (9 + 2 * ((n * m).bit_length() - 1))
    assert ctx.rangeproof_verify_batch(n, m, proofs, pl, coms, b"rate") == bytes(nb)
    g = O.Gens(n, m)
    th = min(os.cpu_count() or 1, 16)
    cnt = max(th, min(nb, 64 * th // m if m < 16 else 2 * th))
    t1 = time.perf_counter()
    O.prove_batch(g, vals[:cnt * m], bl[:32 * cnt * m], m, n, b"rate", b"seed", threads=th)
    cpu = cnt / (time.perf_counter() - t1)
    print("prove (n=%d, m=%d) batch %5d: GPU %.2f ms per batch = %.0f proofs/s (one context, host pointers)   CPU oracle prover %.0f proofs/s on %d threads"
          % (n, m, nb, dt * 1e3, nb / dt, cpu, th), flush=True)
    ctx.close()

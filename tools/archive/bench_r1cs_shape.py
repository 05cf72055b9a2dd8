#!/usr/bin/env python3
"""Dev tool (GPU box): BASELINE config 5 -- the R1CS verifier's MSM SHAPE (r1cs/verifier.rs:459-491, k = 1024 shuffle):
N = 6179 terms = 4098 shared generator terms (BulletproofGens::new(2048, 1) + Pedersen) + 2081 per-proof points with
uniform scalars.  Times bpgpu_msm_batch_shared (PCIe-inclusive host entry point) for one MSM and for a batch of 64, and
the plain bpgpu_msm_batch on the same terms.  No R1CS front end exists here (out of scope); parity of this shape is
tests/test_gpu_msm.py::test_r1cs_shape_msm_config5."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bulletproofs_amd as bp

n, m, nu = 2048, 1, 2081
ngen = 2 * n * m + 2
rng = np.random.default_rng(5)
def scalars(cnt):
    x = rng.integers(0, 256, size=(cnt, 32), dtype=np.uint8); x[:, 31] &= 0x0f
    return x.tobytes()
c = bp.Context(0)
t0 = time.time(); c.gens_create(n, m); print("gens_create(2048, 1): %.2f s, W = %d, tables %.1f GB" % (
    time.time() - t0, c.get_option("fixed_window_bits"), c.get_option("fixed_table_bytes") / 1e9))
G, H, B, Bb = c.gens_export()
pool = (G + H)[:32 * 4096]                      # valid points: reuse generator encodings as the per-proof points
for nb in (1, 64):
    UP = b"".join(pool[32 * ((b * 131 + u) % 4096):32 * ((b * 131 + u) % 4096) + 32] for b in range(nb) for u in range(nu))
    GS, US = scalars(nb * ngen), scalars(nb * nu)
    c.msm_batch_shared(n, m, nb, nu, GS, US, UP)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps): out, st = c.msm_batch_shared(n, m, nb, nu, GS, US, UP)
    dt = (time.perf_counter() - t0) / reps
    assert st == bytes(nb)
    print("msm_batch_shared  batch %2d: %8.3f ms/call  %8.1f MSMs/s  %6.2f M terms/s" % (nb, dt * 1e3, nb / dt, nb * 6179 / dt / 1e6))
    gens_pts = Bb + B + G + H
    flat_s = b"".join(GS[32 * ngen * b:32 * ngen * (b + 1)] + US[32 * nu * b:32 * nu * (b + 1)] for b in range(nb))
    flat_p = b"".join(gens_pts + UP[32 * nu * b:32 * nu * (b + 1)] for b in range(nb))
    c.msm_batch([6179] * nb, flat_s, flat_p)
    t0 = time.perf_counter()
    for _ in range(reps): out2, st2 = c.msm_batch([6179] * nb, flat_s, flat_p)
    dt = (time.perf_counter() - t0) / reps
    assert out2 == out
    print("msm_batch (plain) batch %2d: %8.3f ms/call  %8.1f MSMs/s  %6.2f M terms/s" % (nb, dt * 1e3, nb / dt, nb * 6179 / dt / 1e6))
c.close()

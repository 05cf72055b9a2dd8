#!/usr/bin/env python3
"""Dev tool: times the shared-generator MSM pipeline at a rangeproof shape with per-kernel
HIP-event timing.  Uses the oracle only to manufacture valid random points (dev tool, not bench)."""
import argparse, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import bulletproofs_amd as bp
import pyoracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=64); ap.add_argument("--m", type=int, default=1)
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--W", type=int, default=8)
ap.add_argument("--splits", type=int, default=0); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--check", type=int, default=2)
ap.add_argument("--same-scalars", action="store_true", help="every proof uses the same generator scalars: all lanes gather the same table line")
a = ap.parse_args()
n, m, nb = a.n, a.m, a.batch
k = (n * m).bit_length() - 1
nu = 4 + 2 * k + m
ngen = 2 * n * m + 2
L = 2**252 + 27742317777372353535851937790883648493
rng = np.random.default_rng(1)
def scalars(cnt):
    x = rng.integers(0, 256, size=(cnt, 32), dtype=np.uint8); x[:, 31] &= 0x0f   # < 2^252 < l
    return x.tobytes()
g = O.Gens(n, m); G, H, B, Bb = g.export()
c = bp.Context(0, fixed_window_bits=a.W, fixed_splits=a.splits)
t0 = time.time(); c.gens_load(n, m, G, H, B, Bb); print("gens_load %.3fs (W=%d)" % (time.time() - t0, a.W))
import ctypes as C
out = C.create_string_buffer(32)
upool = []
for i in range(64):
    O.lib().oracle_from_uniform_bytes(hashlib.shake_256(b"u%d" % i).digest(64), out); upool.append(out.raw)
UP = b"".join(upool[(b * 7 + u) % 64] for b in range(nb) for u in range(nu))
GS, US = scalars(nb * ngen), scalars(nb * nu)
if a.same_scalars:
    GS = GS[:32 * ngen] * nb
c.profile_enable(True)
res, st = c.msm_batch_shared(n, m, nb, nu, GS, US, UP)   # warm-up (arena growth)
c.profile_reset()
t0 = time.time()
for _ in range(a.reps):
    res, st = c.msm_batch_shared(n, m, nb, nu, GS, US, UP)
dt = (time.time() - t0) / a.reps
print("n=%d m=%d batch=%d N=%d  host-API wall %.3f ms/batch -> %.0f MSM/s" % (n, m, nb, ngen + nu, dt * 1e3, nb / dt))
rep = c.profile_report(); tot = 0
for name, (cnt, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print("  %-14s %4d launches  %9.3f ms avg" % (name, cnt, ms / cnt)); tot += ms / a.reps
print("  kernels total %.3f ms/batch -> %.0f MSM/s (kernel-only)" % (tot, nb / (tot * 1e-3)))
gp = Bb + B + G + H
for b in range(min(a.check, nb)):
    exp = O.msm(GS[32 * ngen * b:32 * ngen * (b + 1)] + US[32 * nu * b:32 * nu * (b + 1)], gp + UP[32 * nu * b:32 * nu * (b + 1)])
    assert st[b] == 0 and res[32 * b:32 * b + 32] == exp[1], b
print("parity ok on %d" % min(a.check, nb))

#!/usr/bin/env python3
"""Dev tool (GPU box): bpgpu_msm_batch_dev timings, table-lookup path (msm_vb.h) vs bucket path (bucket.h), for a sweep of
terms-per-MSM and batch sizes; inputs resident in HBM, one stream.  Decides BK_MIN_TERMS and the c = 8 / 12 switch."""
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bulletproofs_amd as bp

dev = torch.device("cuda", 0)
L = bp.lib()
base = bp.Context(0)
base.gens_create(64, 8)
G, H, B, Bb = base.gens_export()
gens = [G[32 * i:32 * i + 32] for i in range(512)] + [H[32 * i:32 * i + 32] for i in range(512)]


def inputs(n, nb):
    raw = bytearray(hashlib.shake_256(b"xo-%d-%d" % (n, nb)).digest(32 * n * nb))
    for i in range(31, len(raw), 32):
        raw[i] &= 0x0f
    pts = b"".join(gens[(5 * i + 11 * b) % len(gens)] for b in range(nb) for i in range(n))
    to_dev = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev)
    return to_dev(bytes(raw)), to_dev(pts)


def run(ctx, n, nb, d_s, d_p, reps):
    nt = (C.c_uint32 * nb)(*[n] * nb)
    d_o = torch.zeros((nb, 32), dtype=torch.uint8, device=dev)
    d_t = torch.zeros((nb,), dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(device=dev)
    for _ in range(2):
        assert L.bpgpu_msm_batch_dev(ctx.h, nb, nt, d_s.data_ptr(), d_p.data_ptr(), d_o.data_ptr(), d_t.data_ptr(), s.cuda_stream) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        L.bpgpu_msm_batch_dev(ctx.h, nb, nt, d_s.data_ptr(), d_p.data_ptr(), d_o.data_ptr(), d_t.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, d_o.cpu().numpy().tobytes()


for nb in (1, 64):
    for n in (128, 192, 256, 512, 1024, 2081, 4096, 8192, 20000, 65536):
        if n * nb > 600000:
            continue
        d_s, d_p = inputs(n, nb)
        res = {}
        for name, bm, cbits in (("lookup", 2**31 - 1, 0), ("bucket", 1, 0)):
            ctx = bp.Context(0)
            ctx.set_option("bucket_min_terms", bm)
            res[name] = run(ctx, n, nb, d_s, d_p, 6 if n * nb < 100000 else 3)
            ctx.close()
        assert res["lookup"][1] == res["bucket"][1], (n, nb)
        print("batch %3d x %6d terms: lookup %8.3f ms   bucket %8.3f ms   (%.2fx)" % (nb, n, res["lookup"][0] * 1e3, res["bucket"][0] * 1e3,
                                                                                   res["lookup"][0] / res["bucket"][0]), flush=True)

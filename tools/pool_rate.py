#!/usr/bin/env python3
"""Rates of the library's scheduler (bpgpu_pool_*, include/bpgpu.h) on cfg2 proofs:
  host  : ONE bpgpu_pool_rangeproof_verify call from ONE host thread, host memory in / verdicts out, nbatch = 1024 .. 65536
  burst : K = 20 device-resident batches of 1024 submitted back to back to an idle pool, one flush, wait -- for several
          values of the coalescing width (1024 = one chain per batch, i.e. no coalescing)
  steady: 2048 batches of 1024 through submit_dev with automatic flushes
Every verdict is checked against the planted pattern (three invalid proofs per 1024).
    python tools/pool_rate.py [host] [burst] [steady]"""
import hashlib
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch  # noqa: E402
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd import workload as wl  # noqa: E402


def planted(fx, nb, first=0):
    proofs, coms = wl.tile_batch(fx, nb, first=first)
    pb = bytearray(proofs)
    exp = bytearray(nb)
    for j in range(0, nb, 1024):
        for i in (j + 11, j + 512, min(nb - 1, j + 1023)):
            if i < nb and not exp[i]:
                pb[i * fx.proof_len + 128] ^= 1
                exp[i] = 1
    return bytes(pb), coms, bytes(exp)


def host_rates(fx, sizes=(1024, 4096, 16384, 65536), lanes=32, **opts):
    pool = bp.Pool((0,), lanes, **opts)
    pool.gens_create(fx.n, fx.m)
    for nb in sizes:
        proofs, coms, exp = planted(fx, nb)
        rng = hashlib.shake_256(b"pr").digest(64 * nb)
        ts = []
        for r in range(7):
            t0 = time.perf_counter()
            v = pool.rangeproof_verify(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng)
            ts.append(time.perf_counter() - t0)
            assert v == exp
        t = statistics.median(ts[1:])
        print("host one-call nbatch=%6d lanes=%d opts=%s: %.3f ms  %.2f M verifications/s (best %.2f)  reps ms: %s" % (nb, lanes, opts, t * 1e3, nb / t / 1e6, nb / min(ts) / 1e6, " ".join("%.2f" % (x * 1e3) for x in ts)), flush=True)
    pool.close()


def dev_setup(fx, total):
    dev = torch.device("cuda", 0)
    proofs, coms, exp = planted(fx, total)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    return dev, to_dev(proofs), to_dev(coms), to_dev(hashlib.shake_256(b"pd").digest(64 * total)), torch.frombuffer(bytearray(exp), dtype=torch.uint8).to(dev)


def burst_rates(fx, K=20, nb=1024, widths=(1024, 2048, 4096, 5120, 6912, 10240, 20480), lanes=32, reps=40, **opts):
    dev, d_p, d_c, d_r, d_e = dev_setup(fx, 8 * nb)
    for w in widths:
        pool = bp.Pool((0,), lanes, **opts)
        pool.set_option("coalesce_proofs", w)
        pool.set_option("max_chain_proofs", max(w, 16384))
        pool.set_option("auto_flush_items", 100000)
        pool.gens_create(fx.n, fx.m)
        d_v = torch.full((K, nb), 255, dtype=torch.uint8, device=dev)
        ts = []
        for r in range(reps + 3):
            d_v.fill_(255)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                j = (k + r) % 8
                pool.submit_dev(0, fx.n, fx.m, nb, d_p.data_ptr() + j * nb * fx.proof_len, fx.proof_len, d_c.data_ptr() + j * nb * 32, fx.label,
                                d_r.data_ptr() + j * nb * 64, d_v[k].data_ptr())
            pool.flush()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            for k in range(K):
                j = (k + r) % 8
                assert bool((d_v[k] == d_e[j * nb:(j + 1) * nb]).all().item())
        t = statistics.median(ts[3:])
        print("burst %d x %d coalesce_proofs=%5d %s: %.3f ms  %.2f M/s (min %.3f ms)" % (K, nb, w, opts, t * 1e3, K * nb / t / 1e6, min(ts) * 1e3), flush=True)
        pool.close()


def steady_rates(fx, nb=1024, steps=2048, configs=((128, 1024, 128), (128, 2048, 128), (64, 4096, 128), (32, 4096, 128), (32, 8192, 256))):
    dev, d_p, d_c, d_r, d_e = dev_setup(fx, 8 * nb)
    for lanes, w, af in configs:
        pool = bp.Pool((0,), lanes)
        pool.set_option("coalesce_proofs", w)
        pool.set_option("auto_flush_items", af)
        pool.gens_create(fx.n, fx.m)
        d_v = torch.full((steps, nb), 255, dtype=torch.uint8, device=dev)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                j = k % 8
                pool.submit_dev(0, fx.n, fx.m, nb, d_p.data_ptr() + j * nb * fx.proof_len, fx.proof_len, d_c.data_ptr() + j * nb * 32, fx.label,
                                d_r.data_ptr() + j * nb * 64, d_v[k].data_ptr())
            pool.flush()
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
        ok = all(bool((d_v[k] == d_e[(k % 8) * nb:(k % 8 + 1) * nb]).all().item()) for k in range(0, steps, 37))
        print("steady %d x %d lanes=%d coalesce_proofs=%d auto_flush=%d: %.2f M/s  verdicts %s" % (steps, nb, lanes, w, af, steps * nb / t / 1e6, "ok" if ok else "WRONG"), flush=True)
        pool.close()


if __name__ == "__main__":
    what = sys.argv[1:] or ["host", "burst", "steady"]
    fx = wl.load_fixture("cfg2_n64_m1")
    if "burst" in what:
        burst_rates(fx)
    if "burst2" in what:
        for sp in (16, 32, 64):
            burst_rates(fx, widths=(4096, 5120, 10240, 20480), fixed_splits=sp)
        burst_rates(fx, widths=(2048, 4096, 5120), horner_lanes=64)
        burst_rates(fx, widths=(4096,), K=16)
        burst_rates(fx, widths=(4096,), K=64)
    if "host" in what:
        host_rates(fx)
        host_rates(fx, sizes=(16384, 65536), slice_proofs=1024)
        host_rates(fx, sizes=(16384, 65536), slice_proofs=2048)
        host_rates(fx, sizes=(16384, 65536), slice_proofs=4096, host_workers=8)
    if "steady" in what:
        steady_rates(fx)

#!/usr/bin/env python3
"""Throughput of the HOST-pointer entry point bpgpu_rangeproof_verify_batch (what the Rust drop-in of INTEGRATION.md binds):
PCIe-inclusive, proof bytes in host memory -> verdict bytes in host memory.  T host threads, one context each (ctypes
releases the GIL during the call), each thread verifies different 1024-slices of the cfg2 fixture and checks the verdicts
against the planted pattern.  Never bench.py's `value` (that one has inputs resident in HBM).

    python tools/host_api_rate.py [--threads 1,8,32,64] [--batch 1024] [--calls 40] [--blocking 0|1]
"""
import argparse
import hashlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,8,32,64")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--calls", type=int, default=40, help="calls per thread")
    ap.add_argument("--blocking", type=int, default=-1, help="host_sync_blocking option (default: try both)")
    ap.add_argument("--pipelined", default="", help="comma list of context counts: ONE host thread cycling over that many contexts with "
                                                   "bpgpu_rangeproof_verify_batch_submit / bpgpu_ctx_collect")
    a = ap.parse_args()
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    sys.path.insert(0, ROOT)
    from bench import plant_invalid
    fx = wl.load_fixture("cfg2_n64_m1")
    nsl = fx.count // a.batch
    proofs, coms = wl.tile_batch(fx, nsl * a.batch)
    planted, expect = plant_invalid(proofs, fx.proof_len, a.batch, nsl)
    sl = [(planted[j * a.batch * fx.proof_len:(j + 1) * a.batch * fx.proof_len], coms[j * a.batch * 32:(j + 1) * a.batch * 32], bytes(expect[j]))
          for j in range(nsl)]
    rng = hashlib.shake_256(b"host-rate").digest(64 * a.batch)
    for nctx in [int(x) for x in a.pipelined.split(",") if x]:
        ctxs = []
        for _ in range(nctx):
            c = bp.Context(0)
            c.gens_create(64, 1)
            ctxs.append(c)
        for phase, rounds in (("warm", 2), ("timed", a.calls)):
            t0 = time.perf_counter()
            inflight = [None] * nctx
            done = 0
            for i in range(rounds * nctx):
                k = i % nctx
                if inflight[k] is not None:
                    assert ctxs[k].collect() == inflight[k]
                    done += 1
                p, cm, e = sl[i % nsl]
                ctxs[k].rangeproof_verify_batch_submit(fx.n, fx.m, p, fx.proof_len, cm, fx.label, rng)
                inflight[k] = e
            for k in range(nctx):
                if inflight[k] is not None:
                    assert ctxs[k].collect() == inflight[k]
            dt = time.perf_counter() - t0
        print("host-pointer entry point, ONE thread pipelining %3d contexts (submit/collect) x %d calls of batch %d: %.3f M verifications/s"
              % (nctx, a.calls, a.batch, rounds * nctx * a.batch / dt / 1e6), flush=True)
        for c in ctxs:
            c.close()
    if a.pipelined and a.threads == "0":
        return
    for blocking in ([0, 1] if a.blocking < 0 else [a.blocking]):
        for T in [int(x) for x in a.threads.split(",")]:
            ctxs = []
            for _ in range(T):
                c = bp.Context(0)
                c.set_option("host_sync_blocking", blocking)
                c.gens_create(64, 1)
                ctxs.append(c)
            bad = []

            def work(k, calls):
                for i in range(calls):
                    p, cm, e = sl[(k + i) % nsl]
                    v = ctxs[k].rangeproof_verify_batch(fx.n, fx.m, p, fx.proof_len, cm, fx.label, rng)
                    if v != e:
                        bad.append((k, i))
            ths = [threading.Thread(target=work, args=(k, 2)) for k in range(T)]   # set-up calls
            [t.start() for t in ths]
            [t.join() for t in ths]
            t0 = time.perf_counter()
            ths = [threading.Thread(target=work, args=(k, a.calls)) for k in range(T)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            dt = time.perf_counter() - t0
            assert not bad, bad[:4]
            print("host-pointer entry point: %3d threads x %d calls of batch %d, sync=%s: %.3f M verifications/s (%.0f us per call per thread)"
                  % (T, a.calls, a.batch, "blocking" if blocking else "spin", T * a.calls * a.batch / dt / 1e6, dt / a.calls * 1e6), flush=True)
            for c in ctxs:
                c.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Dev tool (GPU box): throughput of bpgpu_linear_verify_batch / bpgpu_linear_create_batch (host pointers, one context, then
4 contexts on 4 threads) at the sizes of the reference's LinearProof tests, next to the oracle on one host core."""
import hashlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import bulletproofs_amd as bp
import pyoracle as O
ELL = 2 ** 252 + 27742317777372353535851937790883648493
SHAPES = ((16, 4096), (64, 1024), (64, 4096), (256, 1024))
if len(sys.argv) == 3:          # one shape only: `linear_rate.py 64 4096` (e.g. under rocprofv3)
    SHAPES = ((int(sys.argv[1]), int(sys.argv[2])),)
for n, nb in SHAPES:
    base = O.linear_test_instance(n, b"rate%d" % n)
    lg = n.bit_length() - 1
    stream = hashlib.shake_256(b"rate-in%d" % n).digest(64 * (n + 1) * 64)
    red = lambda i: (int.from_bytes(stream[64 * i:64 * i + 64], "little") % ELL).to_bytes(32, "little")
    # 64 distinct witnesses, tiled over the batch; commitments from the oracle's MSM
    As, rs, Cs = [], [], []
    for j in range(64):
        a = b"".join(red(j * (n + 1) + i) for i in range(n))
        r = red(j * (n + 1) + n)
        c = sum(int.from_bytes(a[32 * i:32 * i + 32], "little") * int.from_bytes(base["b"][32 * i:32 * i + 32], "little") for i in range(n)) % ELL
        As.append(a); rs.append(r); Cs.append(O.msm(a + r + c.to_bytes(32, "little"), base["G"] + base["B"] + base["F"])[1])
    A, R, CC = (b"".join(x[j % 64] for j in range(nb)) for x in (As, rs, Cs))
    rng = hashlib.shake_256(b"rng").digest(64 * (2 * lg + 2) * nb)
    ctxs = [bp.Context(0) for _ in range(4)]
    c0 = ctxs[0]
    proofs, status = c0.linear_create_batch(n, CC, R, A, base["b"], base["G"], base["F"], base["B"], label=b"rate", rng=rng)
    pl = len(proofs) // nb
    assert status == bytes(nb) and c0.linear_verify_batch(n, proofs, pl, CC, base["G"], base["F"], base["B"], base["b"], label=b"rate") == bytes(nb)
    def run(fn, reps=3):
        fn(c0)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(c0)
        one = (time.perf_counter() - t0) / reps
        for c_ in ctxs:
            fn(c_)
        ths = [threading.Thread(target=lambda c_=c_: [fn(c_) for _ in range(reps)]) for c_ in ctxs]
        t0 = time.perf_counter()
        [t.start() for t in ths]; [t.join() for t in ths]
        four = (time.perf_counter() - t0) / reps / len(ctxs)
        return one, four
    v1, v4 = run(lambda c_: c_.linear_verify_batch(n, proofs, pl, CC, base["G"], base["F"], base["B"], base["b"], label=b"rate"))
    for c_ in ctxs:
        c_.gens_create(n, 1)
    assert c0.linear_verify_batch(n, proofs, pl, CC, None, None, None, base["b"], label=b"rate") == bytes(nb)
    f1, f4 = run(lambda c_: c_.linear_verify_batch(n, proofs, pl, CC, None, None, None, base["b"], label=b"rate"))
    assert c0.linear_create_batch(n, CC, R, A, base["b"], None, None, None, label=b"rate", rng=rng)[0] == proofs
    q1, q4 = run(lambda c_: c_.linear_create_batch(n, CC, R, A, base["b"], None, None, None, label=b"rate", rng=rng))
    print("LinearProof n=%3d batch %5d: create, bases = the context's generators (window tables): %.2f ms = %.0f proofs/s (1 context), %.0f/s (4 contexts)" % (n, nb, q1 * 1e3, nb / q1, nb / q4))
    p1, p4 = run(lambda c_: c_.linear_create_batch(n, CC, R, A, base["b"], base["G"], base["F"], base["B"], label=b"rate", rng=rng))
    st = O.transcript_new(b"rate")
    cnt = max(4, 2048 // n)
    t0 = time.perf_counter()
    for j in range(cnt):
        assert O.linear_verify(n, proofs[pl * j:pl * (j + 1)], st, CC[32 * j:32 * j + 32], base["G"], base["F"], base["B"], base["b"])[0] == 0
    cv = cnt / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for j in range(cnt):
        O.linear_create(n, st, rng[64 * (2 * lg + 2) * j:], CC[32 * j:32 * j + 32], R[32 * j:32 * j + 32], A[32 * n * j:32 * n * (j + 1)], base["b"], base["G"], base["F"], base["B"])
    cp = cnt / (time.perf_counter() - t0)
    print("LinearProof n=%3d batch %5d: verify, bases = the context's generators (window tables): %.2f ms = %.0f proofs/s (1 context), %.0f/s (4 contexts)" % (n, nb, f1 * 1e3, nb / f1, nb / f4))
    print("LinearProof n=%3d batch %5d: verify %.2f ms = %.0f proofs/s (1 context), %.0f/s (4 contexts)   create %.2f ms = %.0f proofs/s, %.0f/s (4 contexts)"
          "   CPU oracle, one core: verify %.0f/s, create %.0f/s" % (n, nb, v1 * 1e3, nb / v1, nb / v4, p1 * 1e3, nb / p1, nb / p4, cv, cp), flush=True)
    for c_ in ctxs:
        c_.close()

"""Many batches in flight: several contexts (one per HIP stream) share the device and the generator tables, calls
are enqueued round-robin without synchronising in between (what bench.py and a verification service do), per-proof
and batch-combined entry points mixed.  Every verdict byte must equal the oracle's, in every round -- this is the
test that would catch a race on the shared tables, on a context's arena or on its hand-back-clean status words."""
import hashlib
import os

import pytest

pytestmark = pytest.mark.gpu


def test_interleaved_streams_give_oracle_verdicts(oracle, oracle_gens_64_8):
    import torch
    import bulletproofs_amd as bp
    L = bp.lib()
    dev = torch.device("cuda", 0)
    n, m, nb, nctx, rounds = 64, 1, 96, 6, 8
    vals = [int.from_bytes(hashlib.shake_256(b"cv%d" % i).digest(8), "little") for i in range(nb)]
    bl = b"".join(hashlib.shake_256(b"cb%d" % i).digest(31) + b"\x00" for i in range(nb))
    proofs, coms = oracle.prove_batch(oracle_gens_64_8, vals, bl, m, n, b"conc", b"seed", threads=os.cpu_count() or 1)
    pl = oracle.proof_len(n, m)
    ctxs, streams, inputs, expect = [], [], [], []
    for k in range(nctx):
        c = bp.Context(0)
        c.gens_create(64, 8)                       # same generators in every context: one shared table
        ctxs.append(c)
        streams.append(torch.cuda.Stream(device=dev))
        pb = bytearray(proofs)
        for i in range(k, nb, 5 + k):              # a different tampering pattern per context
            pb[i * pl + (i * 31 + 7 * k) % pl] ^= 1 << (k % 8)
        if k == 2:
            pb = bytearray(proofs)                 # one context with a clean batch (the combined check passes)
        rng = hashlib.shake_256(b"conc-rng%d" % k).digest(64 * nb)
        wts = hashlib.shake_256(b"conc-wts%d" % k).digest(64 * nb)
        _, ev, _ = oracle.verify_batch(oracle_gens_64_8, bytes(pb), coms, m, n, b"conc", rng, threads=os.cpu_count() or 1)
        expect.append(ev)
        to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        inputs.append((to_dev(bytes(pb)), to_dev(coms), to_dev(rng), to_dev(wts)))
    assert sum(1 for e in expect if any(e)) >= 4 and not any(expect[2])
    out = torch.full((rounds, nctx, nb), 255, dtype=torch.uint8, device=dev)
    bo = torch.full((rounds, nctx, 33), 255, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for r in range(rounds):
        for k in range(nctx):
            d_p, d_c, d_r, d_w = inputs[k]
            if (r + k) % 3 == 2:                   # batch-combined entry point (no fallback in the _dev variant)
                rc = L.bpgpu_rangeproof_verify_rlc_dev(ctxs[k].h, n, m, nb, d_p.data_ptr(), pl, d_c.data_ptr(), b"conc", 4, d_r.data_ptr(),
                                                       d_w.data_ptr(), out[r, k].data_ptr(), bo[r, k].data_ptr(), streams[k].cuda_stream)
            else:
                rc = L.bpgpu_rangeproof_verify_batch_dev(ctxs[k].h, n, m, nb, d_p.data_ptr(), pl, d_c.data_ptr(), b"conc", 4, d_r.data_ptr(),
                                                         out[r, k].data_ptr(), None, streams[k].cuda_stream)
            assert rc == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    gbo = bo.cpu().numpy()
    for r in range(rounds):
        for k in range(nctx):
            e = list(expect[k])
            if (r + k) % 3 == 2:
                # proofs the front end rejects keep their code; the others are 0 when the combination is the identity,
                # UNDECIDED (5) otherwise -- which happens exactly when some accepted proof fails its check.  A proof whose
                # point fails to decode is rejected by the front end (code 1) and does not spoil the combination.
                g = list(got[r, k])
                assert gbo[r, k][0] in (0, 1)
                if gbo[r, k][0] == 0:
                    assert all((x == 0 and y == 0) or (x == y and y != 0) for x, y in zip(g, e)), (r, k)
                else:
                    assert any(x == 5 and y == 1 for x, y in zip(g, e)), (r, k)
                    assert all(x == 5 or (x == y and y != 0) for x, y in zip(g, e)), (r, k)
            else:
                assert list(got[r, k]) == e, (r, k)
    for c in ctxs:
        c.close()


def test_submit_collect_pipelining_one_thread(oracle, oracle_gens_64_8):
    """bpgpu_rangeproof_verify_batch_submit / bpgpu_ctx_collect: one host thread keeps several contexts busy; results are
    delivered by collect() or implicitly by the next call on the context; every verdict equals the oracle's."""
    import bulletproofs_amd as bp
    n, m, nb, nctx = 64, 1, 128, 5
    vals = [int.from_bytes(hashlib.shake_256(b"pv%d" % i).digest(8), "little") for i in range(nb)]
    bl = b"".join(hashlib.shake_256(b"pb%d" % i).digest(31) + b"\x00" for i in range(nb))
    proofs, coms = oracle.prove_batch(oracle_gens_64_8, vals, bl, m, n, b"pipe", b"seed", threads=os.cpu_count() or 1)
    pl = oracle.proof_len(n, m)
    ctxs = []
    for _ in range(nctx):
        c = bp.Context(0)
        c.gens_create(64, 8)
        ctxs.append(c)
    variants = []
    for k in range(7):
        pb = bytearray(proofs)
        for i in range(k, nb, 9 + k):
            pb[i * pl + 128 + k] ^= 1
        rng = hashlib.shake_256(b"pipe-rng%d" % k).digest(64 * nb)
        _, ev, _ = oracle.verify_batch(oracle_gens_64_8, bytes(pb), coms, m, n, b"pipe", rng, threads=os.cpu_count() or 1)
        variants.append((bytes(pb), rng, ev))
    inflight = [None] * nctx
    for i in range(40):
        k = i % nctx
        if inflight[k] is not None:
            assert ctxs[k].collect() == inflight[k], i
        pb, rng, ev = variants[i % 7]
        ctxs[k].rangeproof_verify_batch_submit(n, m, pb, pl, coms, b"pipe", rng)
        inflight[k] = ev
    # implicit delivery: a synchronous call on a context with a submitted call pending first completes that one
    buf0 = ctxs[0]._pending[0]
    v = ctxs[0].rangeproof_verify_batch(n, m, variants[3][0], pl, coms, b"pipe", variants[3][1])
    assert v == variants[3][2] and buf0.raw[:nb] == inflight[0]
    for k in range(1, nctx):
        assert ctxs[k].collect() == inflight[k]
    for c in ctxs:
        c.close()

#!/usr/bin/env python3
"""Dev tool: latency of ONE small verification call (the literal verify_single / verify_multiple replacement) -- host-pointer call and
device-pointer call + synchronize, median of 40, for batches of 1 .. 1024 proofs and the Horner layouts."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bulletproofs_amd as bp
from bulletproofs_amd import workload as wl

fx = wl.load_fixture("cfg2_n64_m1")
dev = torch.device("cuda", 0)
L = bp.lib()
for lanes in (0, 4, 64):
    c = bp.Context(0, fixed_window_bits=int(os.environ.get("W", "16")), horner_lanes=lanes)
    c.gens_create(64, 1)
    for nb in (1, 16, 64, 256, 1024):
        proofs, coms = wl.tile_batch(fx, nb)
        rng = hashlib.shake_256(b"lat").digest(64 * nb)
        d_p = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev)
        d_c = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to(dev)
        d_r = torch.frombuffer(bytearray(rng), dtype=torch.uint8).to(dev)
        d_v = torch.zeros(nb, dtype=torch.uint8, device=dev)
        th, td = [], []
        for i in range(45):
            t0 = time.perf_counter()
            v = c.rangeproof_verify_batch(64, 1, proofs, fx.proof_len, coms, fx.label, rng)
            th.append(time.perf_counter() - t0)
            assert not any(v)
            t0 = time.perf_counter()
            rc = L.bpgpu_rangeproof_verify_batch_dev(c.h, 64, 1, nb, d_p.data_ptr(), fx.proof_len, d_c.data_ptr(), fx.label, len(fx.label), d_r.data_ptr(),
                                                     d_v.data_ptr(), None, None)
            c.synchronize()
            td.append(time.perf_counter() - t0)
            assert rc == 0
        th, td = sorted(th[5:]), sorted(td[5:])
        print("horner_lanes=%-2d nb=%-5d host call %.3f ms   device call + sync %.3f ms" % (lanes, nb, th[len(th) // 2] * 1e3, td[len(td) // 2] * 1e3))
    c.profile_enable(True) if hasattr(c, "profile_enable") else None
    c.close()

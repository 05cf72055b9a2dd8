#!/usr/bin/env python3
"""Writes the input file of `combine_rate msm` (tools/combine_rate.cpp): config 5's 64 multiscalar multiplications
(workload.cfg5_inputs; the per-MSM points are derived on the GPU by a small-table helper context, as bench.py does) followed by the
ORACLE's 32-byte encoding of each (committed: bench_data/cfg5_expected.json, tools/gen_cfg5_expected.py).
    python tools/make_msm_inputs.py out.bin"""
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write(path, device=0):
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    n, nu, nb = wl.CFG5["n"], wl.CFG5["n_unique"], wl.CFG5["nbatch"]
    hc = bp.Context(device, fixed_window_bits=2)
    hc.gens_create(n, 2)
    G2, H2, _, _ = hc.gens_export()
    hc.close()
    gs, us, up = wl.cfg5_inputs(G2, H2, nb)
    with open(os.path.join(ROOT, "bench_data", "cfg5_expected.json")) as f:
        exp = json.load(f)
    with open(path, "wb") as f:
        f.write(struct.pack("<4I", n, 1, nu, nb))
        f.write(gs)
        f.write(us)
        f.write(up)
        f.write(b"".join(bytes.fromhex(exp["msm%d" % b]) for b in range(nb)))
    return path


if __name__ == "__main__":
    print(write(sys.argv[1]))

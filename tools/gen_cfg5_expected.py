#!/usr/bin/env python3
"""Writes bench_data/cfg5_expected.json: the 32-byte encodings the ORACLE computes for every MSM (0 .. 63) of bench.py's
config-5 batch (workload.cfg5_inputs: 4098 generator terms + 2081 from_uniform_bytes points, uniform scalars).  bench.py and
tests/test_gpu_msm.py compare the GPU's results with these committed constants; the oracle itself is not run at bench time.
    python tools/gen_cfg5_expected.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import pyoracle as O
from bulletproofs_amd import workload as wl

n, nu, nb = wl.CFG5["n"], wl.CFG5["n_unique"], wl.CFG5["nbatch"]
ng = 2 * n + 2
G2, H2, B, Bb = O.Gens(n, 2).export()
G1, H1, B1, Bb1 = O.Gens(n, 1).export()
assert G2[:32 * n] == G1 and H2[:32 * n] == H1 and B == B1
gs, us, up = wl.cfg5_inputs(G2, H2, nb)
gens_pts = Bb + B + G1 + H1                                  # the order of the shared terms: B_blinding, B, G, H (mod.rs:433-443)
out = {"_note": "oracle (C restatement, oracle/c) results for MSM 0 .. %d of workload.cfg5_inputs (MSM 0 and %d cross-checked with the oracle's second MSM algorithm); "
               "tools/gen_cfg5_expected.py" % (nb - 1, nb - 1)}
for b in range(nb):
    st, enc = O.msm(gs[32 * ng * b:32 * ng * (b + 1)] + us[32 * nu * b:32 * nu * (b + 1)], gens_pts + up[32 * nu * b:32 * nu * (b + 1)])
    assert st == 0
    if b in (0, nb - 1):
        st2, enc2 = O.msm(gs[32 * ng * b:32 * ng * (b + 1)] + us[32 * nu * b:32 * nu * (b + 1)], gens_pts + up[32 * nu * b:32 * nu * (b + 1)], 1)
        assert enc2 == enc                                   # both MSM algorithms of the oracle agree
    out["msm%d" % b] = enc.hex()
with open(os.path.join(ROOT, "bench_data", "cfg5_expected.json"), "w") as f:
    json.dump(out, f, indent=1)
print({k: out[k] for k in ('msm0', 'msm%d' % (nb - 1))}, len(out) - 1, 'encodings')

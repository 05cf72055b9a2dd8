#!/usr/bin/env python3
"""Dev tool (GPU box): duration of launch 1 (k_rp_stage1) at cfg2, one stream, for library builds with parts of the launch
compiled out (ab/exp_*.so, built with -DBP_EXP_NOTR / NOSC / NOPT / NOKECCAK: timing experiments only -- their verdicts are
meaningless).  Usage: stage1_breakdown.py <variant.so> ...
Building a variant (in bulletproofs_amd/csrc, after a normal build):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBP_EXP_NOSC -DBP_EXP_NOPT -c -o /tmp/k_rp1_onlytr.o k_rp1.hip
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/exp_onlytr.so $(ls build/*.o | grep -v k_rp1.o) /tmp/k_rp1_onlytr.o"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bulletproofs_amd", "csrc", "libbpgpu.so")
SNIP = r'''
import sys; sys.path.insert(0, %r)
import bulletproofs_amd as bp
from bulletproofs_amd import workload as wl
fx = wl.load_fixture("cfg2_n64_m1")
c = bp.Context(0, fixed_window_bits=10)
c.gens_create(64, 1)
nb = 1024
pr, cm = fx.proofs[:nb * fx.proof_len], fx.commitments[:nb * 32]
for _ in range(3): c.rangeproof_verify_batch(64, 1, pr, fx.proof_len, cm, fx.label, bytes(64 * nb))
c.profile_enable(True); c.profile_reset()
for _ in range(20): c.rangeproof_verify_batch(64, 1, pr, fx.proof_len, cm, fx.label, bytes(64 * nb))
r = c.profile_report()
print("%%-14s rp_stage1 %%7.1f us" %% (%r, r["rp_stage1"][1] / r["rp_stage1"][0] * 1e3))
''' % (ROOT, "%s")
keep = LIB + ".keep"
shutil.copy(LIB, keep)
try:
    for v in sys.argv[1:]:
        shutil.copy(v, LIB)
        name = os.path.basename(v).replace("exp_", "").replace(".so", "")
        subprocess.run([sys.executable, "-c", SNIP.replace("%s", name)], check=False)
finally:
    shutil.copy(keep, LIB)
    os.remove(keep)

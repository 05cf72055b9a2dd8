#!/bin/bash
# round 6, call 40: a few small MSMs per call in the narrow form (msm_narrow): parity, single-call latency of the golden MSM sizes A/B
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call40
mkdir -p $OUT
cd $REPO
timeout 2000 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pool_msm.py tests/test_gpu_ipp.py tests/test_gpu_reference_api.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
python - > $OUT/msm_narrow_ab.txt 2>&1 <<'PY'
import hashlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import bulletproofs_amd as bp
import pyoracle as O
import ctypes as C
L = 2**252 + 27742317777372353535851937790883648493
def inputs(n, tag):
    out = C.create_string_buffer(32); S = b""; P = b""
    for i in range(n):
        S += (int.from_bytes(hashlib.shake_256(b"%s-s%d" % (tag, i)).digest(64), "little") % L).to_bytes(32, "little")
        O.lib().oracle_from_uniform_bytes(hashlib.shake_256(b"%s-p%d" % (tag, i)).digest(64), out); P += out.raw
    return S, P
ctxs = {}
for v in (0, 1):
    c = bp.Context(0); c.set_option("msm_narrow", v); ctxs[v] = c
print("one blocking bpgpu_msm_batch call (host pointers in and out), one MSM of N terms: p50 of 300 calls, ms")
for n in (29, 81, 147, 280, 542, 1024):
    S, P = inputs(n, b"ab%d" % n)
    exp = O.msm(S, P)[1]
    row = []
    for rep in range(2):
        for v in (0, 1):
            c = ctxs[v]
            for _ in range(20): out, st = c.msm_batch([n], S, P)
            assert out == exp and st == bytes(1)
            ts = []
            for _ in range(300):
                t0 = time.perf_counter(); c.msm_batch([n], S, P); ts.append(time.perf_counter() - t0)
            ts.sort(); row.append("msm_narrow=%d %.3f" % (v, ts[150] * 1e3))
    print("N = %4d: %s" % (n, "  ".join(row)))
S, P = inputs(8 * 147, b"b8")
for v in (0, 1):
    c = ctxs[v]
    for _ in range(10): c.msm_batch([147] * 8, S, P)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); c.msm_batch([147] * 8, S, P); ts.append(time.perf_counter() - t0)
    ts.sort(); print("8 x 147 terms in one call: msm_narrow=%d p50 %.3f ms" % (v, ts[100] * 1e3))
PY
cat $OUT/msm_narrow_ab.txt

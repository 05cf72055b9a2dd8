#!/bin/bash
# round 6, call 42: bpgpu_msm_batch_shared in the narrow form: parity (test_gpu_msm.py) and single-call latency A/B (the mega-check of one proof as the caller's MSM)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call42
mkdir -p $OUT
cd $REPO
timeout 2000 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pool_msm.py tests/test_gpu_prover_msm.py tests/test_gpu_linear.py tests/test_gpu_audit.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
python - > $OUT/msm_shared_narrow_ab.txt 2>&1 <<'PY'
import hashlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bulletproofs_amd as bp
L = 2**252 + 27742317777372353535851937790883648493
sc = lambda tag: (int.from_bytes(hashlib.shake_256(tag).digest(64), "little") % L).to_bytes(32, "little")
ctxs = {}
for v in (0, 1):
    c = bp.Context(0); c.set_option("msm_narrow", v); c.gens_create(64, 1); ctxs[v] = c
G, H, B, Bb = ctxs[0].gens_export()
print("one blocking bpgpu_msm_batch_shared call (host pointers), n = 64, m = 1 (130 generator terms from the default tables) + n_unique own points: p50 of 300 calls, ms")
for nb, nu in ((1, 17), (1, 40), (4, 17), (16, 17), (1, 300)):
    GS = b"".join(sc(b"g%d-%d" % (nb, i)) for i in range(130 * nb))
    US = b"".join(sc(b"u%d-%d" % (nu, i)) for i in range(nu * nb))
    UP = ((G + H) * 8)[:32 * nu * nb]
    ref = None; row = []
    for rep in range(2):
        for v in (0, 1):
            c = ctxs[v]
            for _ in range(20): res = c.msm_batch_shared(64, 1, nb, nu, GS, US, UP)
            if ref is None: ref = res
            assert res == ref
            ts = []
            for _ in range(300):
                t0 = time.perf_counter(); c.msm_batch_shared(64, 1, nb, nu, GS, US, UP); ts.append(time.perf_counter() - t0)
            ts.sort(); row.append("msm_narrow=%d %.3f" % (v, ts[150] * 1e3))
    print("nbatch %2d, n_unique %3d: %s" % (nb, nu, "  ".join(row)))
PY
cat $OUT/msm_shared_narrow_ab.txt

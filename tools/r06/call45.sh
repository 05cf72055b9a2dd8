#!/bin/bash
# round 6, call 45: four table levels in the narrow small-MSM forms (<= 256 terms) and finish8 for 3-4 partial sums: parity + latency A/B
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call45
mkdir -p $OUT
cd $REPO
timeout 2000 python -m pytest tests/test_gpu_msm.py tests/test_gpu_narrow_chain.py tests/test_gpu_ipp.py tests/test_gpu_pool_msm.py tests/test_gpu_combine.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
python - > $OUT/msm_narrow4_ab.txt 2>&1 <<'PY'
import hashlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bulletproofs_amd as bp
L = 2**252 + 27742317777372353535851937790883648493
sc = lambda tag: (int.from_bytes(hashlib.shake_256(tag).digest(64), "little") % L).to_bytes(32, "little")
ctxs = {}
for v in (0, 256, 768):
    c = bp.Context(0); c.set_option("msm_narrow4_terms", v); c.gens_create(64, 1); ctxs[v] = c
G, H, B, Bb = ctxs[0].gens_export()
def p50(f):
    for _ in range(20): f()
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[150] * 1e3
print("p50 of 300 blocking calls, ms; msm_narrow4_terms = 0 (two table levels) / 256 / 768 (four levels up to that many terms)")
for n in (29, 147, 256, 400):
    S = b"".join(sc(b"s%d-%d" % (n, i)) for i in range(n)); P = ((G + H) * 8)[:32 * n]
    ref = ctxs[0].msm_batch([n], S, P)
    row = []
    for rep in range(2):
        for v in (0, 256, 768):
            assert ctxs[v].msm_batch([n], S, P) == ref
            row.append("%d: %.3f" % (v, p50(lambda: ctxs[v].msm_batch([n], S, P))))
    print("bpgpu_msm_batch, one MSM of %3d terms: %s" % (n, "  ".join(row)))
for nb, nu in ((1, 17), (4, 17), (16, 17), (1, 300)):
    GS = b"".join(sc(b"g%d-%d" % (nb, i)) for i in range(130 * nb)); US = b"".join(sc(b"u%d-%d" % (nu, i)) for i in range(nu * nb)); UP = ((G + H) * 8)[:32 * nu * nb]
    ref = ctxs[0].msm_batch_shared(64, 1, nb, nu, GS, US, UP)
    row = []
    for rep in range(2):
        for v in (0, 256, 768):
            assert ctxs[v].msm_batch_shared(64, 1, nb, nu, GS, US, UP) == ref
            row.append("%d: %.3f" % (v, p50(lambda: ctxs[v].msm_batch_shared(64, 1, nb, nu, GS, US, UP))))
    print("bpgpu_msm_batch_shared, %2d MSMs of 130 table terms + %3d points: %s" % (nb, nu, "  ".join(row)))
PY
cat $OUT/msm_narrow4_ab.txt
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -Wl,-rpath,$LIB -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
for rep in 1 2 3; do for mode in "threads 1" "threads 4" "threads 16" "threads 64"; do /tmp/combine_rate $INP 1.0 $mode 2>/dev/null | grep '^{' | tail -1 | cut -c1-260 >> $OUT/final_rates.txt; done; done
cat $OUT/final_rates.txt

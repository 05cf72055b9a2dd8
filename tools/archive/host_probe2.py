import sys, os, time, hashlib, statistics
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import pool_rate as pr
import bulletproofs_amd as bp
from bulletproofs_amd import workload as wl
fx = wl.load_fixture("cfg2_n64_m1")
print("cpus", len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "-")
for opts in ({}, {"host_workers": 1}, {"host_workers": 4}, {"slice_proofs": 2048}):
    pool = bp.Pool((0,), 16)
    for k, v in opts.items():
        pool.set_option(k, v)
    pool.gens_create(64, 1)
    for nb in (16384, 4096, 65536, 16384):
        proofs, coms, exp = pr.planted(fx, nb)
        rng = hashlib.shake_256(b"pr").digest(64 * nb)
        ts = []
        for r in range(10):
            t0 = time.perf_counter()
            v = pool.rangeproof_verify(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng)
            ts.append(time.perf_counter() - t0)
            assert v == exp
        print(opts, nb, "median %.2f ms" % (statistics.median(ts) * 1e3), " ".join("%.1f" % (x * 1e3) for x in ts), flush=True)
    pool.close()

import sys, os, hashlib, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bulletproofs_amd as bp
dev = torch.device("cuda", 0); L = bp.lib()
n, m, nu = 2048, 1, 2081; ng = 2*n*m+2
for bm in (0, 2**31-1):
    c = bp.Context(0); 
    if bm: c.set_option("bucket_min_terms", bm)
    c.gens_create(n, m)
    G, H, B, Bb = c.gens_export()
    raw = bytearray(hashlib.shake_256(b"s").digest(32*(ng+nu)))
    for i in range(31, len(raw), 32): raw[i] &= 0x0f
    gens = [G[32*i:32*i+32] for i in range(n)]
    up = b"".join(gens[(7*i) % n] for i in range(nu))
    to = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_gs, d_us, d_up = to(bytes(raw[:32*ng])), to(bytes(raw[32*ng:])), to(up)
    d_o = torch.zeros((1,32), dtype=torch.uint8, device=dev); d_t = torch.zeros((1,), dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(device=dev)
    for _ in range(3): L.bpgpu_msm_batch_shared_dev(c.h, n, m, 1, nu, d_gs.data_ptr(), d_us.data_ptr(), d_up.data_ptr(), d_o.data_ptr(), d_t.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    c.profile_reset(); c.profile_enable(True)
    t0=time.perf_counter()
    for _ in range(10): L.bpgpu_msm_batch_shared_dev(c.h, n, m, 1, nu, d_gs.data_ptr(), d_us.data_ptr(), d_up.data_ptr(), d_o.data_ptr(), d_t.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    back_to_back = (time.perf_counter()-t0)/10*1e3
    lat = []
    for _ in range(10):   # one call at a time: the latency of a lone MSM (enqueue + chain + sync)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        L.bpgpu_msm_batch_shared_dev(c.h, n, m, 1, nu, d_gs.data_ptr(), d_us.data_ptr(), d_up.data_ptr(), d_o.data_ptr(), d_t.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize(); lat.append((time.perf_counter() - t1) * 1e3)
    lat.sort()
    print("bucket" if not bm else "lookup", "single MSM: %.3f ms per call back to back on one stream, %.3f ms median latency of a lone call" % (back_to_back, lat[5]), {k: round(v[1]/v[0]*1e3,1) for k,v in sorted(c.profile_report().items(), key=lambda kv:-kv[1][1])})
    c.close()

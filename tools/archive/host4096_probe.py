import sys; sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import pool_rate as pr
from bulletproofs_amd import workload as wl
fx = wl.load_fixture("cfg2_n64_m1")
for opts in ({}, {"slice_proofs": 512}, {"slice_proofs": 2048}, {"slice_proofs": 4096}, {"horner_lanes": 64}, {"horner_lanes": 64, "slice_proofs": 512}, {"horner_lanes": 64, "slice_proofs": 2048},
             {"host_workers": 4}, {"host_workers": 16, "slice_proofs": 256}):
    pr.host_rates(fx, sizes=(4096, 16384), lanes=16, **opts)

#!/usr/bin/env python3
"""Dev tool: one host-pointer pool call of a few thousand proofs against slice width / Horner form (tools/pool_rate.py host_rates)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pool_rate as pr
from bulletproofs_amd import workload as wl
fx = wl.load_fixture("cfg2_n64_m1")
for opts in ({}, {"slice_proofs": 1024}, {"slice_proofs": 1408}, {"slice_proofs": 4096}, {"slice_proofs": 1024, "host_workers": 4}):
    pr.host_rates(fx, sizes=(4096, 8192), lanes=32, **opts)

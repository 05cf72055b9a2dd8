import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pool_rate as pr
from bulletproofs_amd import workload as wl
fx = wl.load_fixture("cfg2_n64_m1")
for hl in (0,):
    pr.host_rates(fx, sizes=(1024, 2048, 4096, 6144, 16384, 65536), lanes=32, horner_lanes=hl)

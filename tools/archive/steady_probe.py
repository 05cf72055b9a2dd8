import sys, os
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import pool_rate as pr
from bulletproofs_amd import workload as wl
fx = wl.load_fixture("cfg2_n64_m1")
cfgs = [(64, 5120, af) for af in (5, 10, 15, 20, 30, 40, 64, 128, 256)] + [(64, 2560, 10), (64, 10240, 20), (64, 10240, 40), (16, 5120, 10), (8, 5120, 5)]
pr.steady_rates(fx, steps=4096, configs=cfgs)

"""The N > 1 plumbing exercised on the GPU box (one MI355X): RCCL process group with one rank (init, per-rank set_device,
all_gather / all_reduce on device tensors), and bench.py's multi-rank path end to end with two ranks sharing cuda:0 (the
shards are verified by the ENGINE; ranks that share a GPU gather over gloo, distinct GPUs use RCCL).  The world_size-2
CPU test of the sharding helpers is tests/test_distributed_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_NCCL_SNIPPET = r"""
import os, sys, hashlib
sys.path.insert(0, %r)
import torch
from bulletproofs_amd import dist as bpdist, workload as wl
import bulletproofs_amd as bp
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r, w = bpdist.init("nccl", dev)
assert (r, w) == (0, 1)
fx = wl.load_fixture("cfg1_n32_m1")
ctx = bp.Context(0, fixed_window_bits=8)
ctx.gens_create(fx.n, fx.m)
proofs, coms = wl.tile_batch(fx, 6)
pb = bytearray(proofs); pb[4 * fx.proof_len + 140] ^= 1
lo, hi = wl.shard_range(6, w, r)
v = ctx.rangeproof_verify_batch(fx.n, fx.m, bytes(pb[lo * fx.proof_len:hi * fx.proof_len]), fx.proof_len, coms[lo * 32:hi * 32], fx.label,
                                hashlib.shake_256(b"d").digest(64 * (hi - lo)))
local = torch.tensor(list(v), dtype=torch.uint8, device=dev)
allv = bpdist.gather_verdicts(local, w)            # a real RCCL all_gather (world of one)
assert allv.is_cuda and allv.shape == (1, 6) and allv.flatten().tolist() == [0, 0, 0, 0, 1, 0]
assert bpdist.max_over_ranks(1.25, w, dev) == 1.25  # RCCL all_reduce(MAX)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("nccl-ok")
"""


def test_rccl_world_of_one_gathers_engine_verdicts():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _NCCL_SNIPPET % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "nccl-ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_runs_with_n_ranks(gpus):
    """`python bench.py --gpus N` launches its own ranks; with more ranks than GPUs they share cuda:0.  bench.py itself checks
    every verdict row against the planted pattern, so a line printed = verdicts correct on every rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "12", "--warmup", "2", "--streams", "4",
                          "--window-bits", "10", "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True, timeout=300)
    if out.returncode != 0:   # keep the ranks' own tracebacks (gpurun_out/ travels back from the GPU box)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "test_bench_ranks_%d.err" % gpus), "w") as f:
            f.write(out.stderr)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == gpus and j["steps"] == 12 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["global_batch"] == 1024 * gpus and j["roofline"]["launches"] >= 12
    # the record describes itself: the ranks the process group really had, its backend (two ranks on one GPU: gloo, so
    # `rccl_ranks` must NOT claim them), every rank's own rate, table-build time and device
    mg = j["multi_gpu"]
    assert mg["ranks_in_process_group"] == gpus and len(mg["per_rank_verifications_per_s"]) == len(mg["per_rank_table_build_s"]) == len(mg["per_rank_device"]) == gpus
    assert all(v > 0 for v in mg["per_rank_verifications_per_s"]) and all(t > 0 for t in mg["per_rank_table_build_s"])
    import torch
    share = gpus > torch.cuda.device_count()          # (on a multi-GPU node the two ranks get a GPU each and the backend is RCCL)
    assert mg["ranks_share_gpus"] == share
    if gpus == 1:
        assert mg["backend"].startswith("none") and mg["rccl_ranks"] == 0
    else:
        assert mg["backend"] == ("gloo" if share else "nccl") and mg["rccl_ranks"] == (0 if share else gpus)

"""world_size-2 gloo test of the N>1 path: contiguous sharding of independent proofs, no data-path
collective, one all_gather of the verdict bytes (bulletproofs_amd/dist.py, used verbatim by bench.py).
On CPU the per-shard verification is done by the oracle (test stand-in for the GPU engine)."""
import hashlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch
    import pyoracle as O
    from bulletproofs_amd import dist as bpdist, workload as wl
    r, w = bpdist.init("gloo")
    assert (r, w) == (rank, world)
    fx = wl.load_fixture("cfg1_n32_m1")
    total = 6
    proofs, coms = wl.tile_batch(fx, total)
    pb = bytearray(proofs)
    pb[4 * fx.proof_len + 140] ^= 1               # global proof 4 is bad
    lo, hi = wl.shard_range(total, world, rank)
    rng = hashlib.shake_256(b"dist").digest(64 * total)
    g = O.Gens(fx.n, fx.m)
    _, v, _ = O.verify_batch(g, bytes(pb[lo * fx.proof_len:hi * fx.proof_len]), coms[lo * 32:hi * 32], fx.m, fx.n, fx.label,
                             rng[64 * lo:64 * hi], threads=1)
    local = torch.tensor(list(v), dtype=torch.uint8)
    allv = bpdist.gather_verdicts(local, world)
    tmax = bpdist.max_over_ranks(1.0 + rank, world)
    q.put((rank, allv.flatten().tolist(), tmax))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_sharded_verify_and_gather_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, allv, tmax in res:
        assert allv == [0, 0, 0, 0, 1, 0] and tmax == 2.0

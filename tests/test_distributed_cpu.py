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
    per_rank = bpdist.gather_floats([100.0 + rank, 0.5 * rank], world)   # (the SCALE record's per-rank figures travel this way)
    assert per_rank == [[100.0 + r_, 0.5 * r_] for r_ in range(world)]
    q.put((rank, allv.flatten().tolist(), tmax))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _free_port():
    """a rendezvous port the kernel just handed out (a pid-derived one can collide on a shared box -- bench.py's self_launch does the same)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_sharded_verify_and_gather_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, allv, tmax in res:
        assert allv == [0, 0, 0, 0, 1, 0] and tmax == 2.0


class _FakeCtx:
    def profile_reset(self):
        pass

    def profile_enable(self, on):
        pass

    def profile_report(self):
        return {}


class _FakeRunner:
    """stands in for bench.py's runner: a region 'takes' a rank-dependent time and ends in one collective, as the real one does"""
    def __init__(self, rank, world):
        self.nstreams, self.ctxs, self.d_verdicts, self.rank, self.world, self.regions = 4, [_FakeCtx()], [0], rank, world, 0
        self.pool = None

    def profile_reset(self):
        pass

    def step(self, k, row):
        pass

    def region(self, K, fence, gather=None):
        from bulletproofs_amd import dist as bpdist
        self.regions += 1
        bpdist.max_over_ranks(0.0, self.world)           # the per-region collective: unequal region counts would deadlock here
        return (0.02 if self.rank == 0 else 0.2), 0.0, None

    def set_profile(self, on, every=1):
        pass

    def kernel_times(self):
        return {}


def _timed_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from bulletproofs_amd import dist as bpdist
    bpdist.init("gloo")
    b = _FakeRunner(rank, world)
    r = bench.timed(b, 12, 0, lambda: None, 0, None, False, True, lambda x: bpdist.max_over_ranks(x, world))
    q.put((rank, len(r["regions"])))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_bench_region_count_agrees_across_ranks_world2():
    """bench.py repeats short timed regions; the repeat count must come from the slowest rank, or ranks whose clocks differ
    run different numbers of collectives and the job deadlocks (seen at --gpus 2 on one box)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] == 5                         # ceil(1 / 0.2) regions on both ranks

"""Synthetic workloads (bench_data/*.bin, written by tools/gen_bench_inputs.py) and the
batch-sharding helpers of the multi-GPU path.  Pure host logic: no oracle, no GPU calls."""
import os
import struct
from collections import namedtuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Fixture = namedtuple("Fixture", "name n m count proof_len label proofs commitments")

# BASELINE.json configs -> fixture and per-GPU batch
CONFIGS = {
    "cfg1": ("cfg1_n32_m1", 1),        # single 32-bit proof (CPU-runnable plumbing case)
    "cfg2": ("cfg2_n64_m1", 1024),     # 1024 single 64-bit proofs on one GPU (the headline config)
    "cfg3": ("cfg3_n64_m16", 256),     # m = 16 aggregated, batch 256
    "cfg4": ("cfg4_n64_m32", 512),     # m = 32 aggregated, 4096 over 8 GPUs = 512 per GPU
}


def load_fixture(name):
    path = os.path.join(ROOT, "bench_data", name + ".bin")
    with open(path, "rb") as f:
        blob = f.read()
    assert blob[:8] == b"BPBENCH1", "bad fixture " + path
    n, m, count, proof_len, label_len = struct.unpack_from("<IIIII", blob, 8)
    off = 28
    label = blob[off:off + label_len]
    off += label_len
    rec = proof_len + 32 * m
    import numpy as np
    recs = np.frombuffer(blob, dtype=np.uint8, count=rec * count, offset=off).reshape(count, rec)
    return Fixture(name, n, m, count, proof_len, label, recs[:, :proof_len].tobytes(), recs[:, proof_len:].tobytes())


def tile_batch(fx, batch, first=0):
    """`batch` proofs starting at logical index `first`, cycling through the fixture's distinct proofs."""
    import numpy as np
    idx = (first + np.arange(batch)) % fx.count
    p = np.frombuffer(fx.proofs, dtype=np.uint8).reshape(fx.count, fx.proof_len)[idx]
    c = np.frombuffer(fx.commitments, dtype=np.uint8).reshape(fx.count, 32 * fx.m)[idx]
    return p.tobytes(), c.tobytes()


def shard_range(total, world_size, rank):
    """Contiguous shard [lo, hi) of `total` independent proofs for `rank`."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def msm_terms(n, m):
    return 2 * n * m + 2 * ((n * m).bit_length() - 1) + m + 6


def reference_point_ops(N):
    """A(N): additions+doublings of the reference's own MSM algorithm."""
    if N < 190:
        return int(N * (256 / 6 + 8) + 256)
    w, cols = (6, 43) if N < 500 else ((7, 37) if N < 800 else (8, 33))
    return cols * (N + 2 ** w - 2) + 256


def algorithmic_bytes_per_verification(n, m):
    """MSM-boundary figure of SURVEY.md section 8(d): 32N scalars + 32(4+2k+m) unique points + 32 result."""
    k = (n * m).bit_length() - 1
    return 32 * msm_terms(n, m) + 32 * (4 + 2 * k + m) + 32

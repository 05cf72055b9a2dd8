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


def executed_point_ops(n, m, window_bits, nsplit, wide=True):
    """Point additions + doublings this engine performs per verification: one mixed addition per (generator,
    window) of the table walk, `nsplit` additions to fold the per-split partial sums, per unique point 7 operations for its
    {1..8}P table and one addition in each of the 64 radix-16 windows, and one Horner chain of 252 doublings + 64 additions.
    wide (chains of >= 2048 proofs, what the pool issues): A -- coefficient 1 -- has no table and no window sums, it is one addition
    after the chain; narrow chains: A like every other point, plus the column sums over chunks of 32 points when there are more."""
    k = (n * m).bit_length() - 1
    U = 4 + 2 * k + m
    nwin = -(-255 // window_bits)
    walk = (2 * n * m + 2) * nwin + nsplit
    if wide:
        return walk + (U - 1) * (7 + 64) + 1 + 252 + 64
    chunks = -(-U // 32)
    return walk + U * (7 + 64) + (chunks - 1) * 64 + 252 + 63


def algorithmic_bytes_per_verification(n, m):
    """MSM-boundary figure of SURVEY.md section 8(d): 32N scalars + 32(4+2k+m) unique points + 32 result."""
    k = (n * m).bit_length() - 1
    return 32 * msm_terms(n, m) + 32 * (4 + 2 * k + m) + 32


# ---- BASELINE config 5: the R1CS verifier's MSM shape ------------------------------------------------
CFG5 = dict(n=2048, m=1, n_unique=2081, nbatch=64)


def cfg5_inputs(G2, H2, nbatch=64):
    """Inputs of the config-5 MSM batch: uniform scalars (SHAKE256-derived, top nibble cleared: canonical) for the 4098 generator
    terms and the 2081 per-MSM points of each MSM, and the per-MSM points themselves: RistrettoPoint::from_uniform_bytes outputs,
    as SURVEY.md 8d specifies -- the party-1 chains of BulletproofGens::new(2048, 2) (SHAKE256("GeneratorsChain" || 'G'/'H' ||
    u32le(1)) -> from_uniform_bytes, generators.rs:58-104), 4096 points none of which is among the MSM's own generator terms
    (party 0), rotated by 31 per MSM.  G2 / H2: the exported G / H encodings of a (2048, 2) generator set.
    Returns (gen_scalars, uniq_scalars, uniq_points) as bytes."""
    import hashlib
    n, nu = CFG5["n"], CFG5["n_unique"]
    ng = 2 * n + 2
    raw = bytearray(hashlib.shake_256(b"cfg5-scalars").digest(32 * (ng + nu) * nbatch))
    for i in range(31, len(raw), 32):
        raw[i] &= 0x0f
    assert len(G2) == len(H2) == 32 * 2 * n
    pool = G2[32 * n:] + H2[32 * n:]                  # party 1
    upts = b"".join(pool[32 * ((i + 31 * b) % (2 * n)):32 * ((i + 31 * b) % (2 * n)) + 32] for b in range(nbatch) for i in range(nu))
    return bytes(raw[:32 * ng * nbatch]), bytes(raw[32 * ng * nbatch:]), upts

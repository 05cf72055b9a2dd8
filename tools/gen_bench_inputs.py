#!/usr/bin/env python3
"""Generates the synthetic range proofs bench.py verifies.

The proofs are made by the oracle's (non-constant-time) prover from SHAKE256-derived values and
blindings, so the files are reproducible:   python tools/gen_bench_inputs.py
File format (little-endian): magic "BPBENCH1", u32 n, u32 m, u32 count, u32 proof_len, u32 label_len,
label bytes, then count * (proof_len proof bytes + m*32 commitment bytes).
This script is the only place where bench inputs touch oracle/; bench.py just reads the files.
"""
import hashlib, os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle")]
import pyoracle as O

CONFIGS = [  # name, n, m, distinct proofs
    ("cfg2_n64_m1", 64, 1, 8192),
    ("cfg3_n64_m16", 64, 16, 256),
    ("cfg4_n64_m32", 64, 32, 512),
    ("cfg1_n32_m1", 32, 1, 4),
]
LABEL = b"AggregateRangeProofBenchmark"   # benches/range_proof.rs:80 of the reference

def main():
    threads = os.cpu_count() or 1
    for name, n, m, count in CONFIGS:
        path = os.path.join(ROOT, "bench_data", name + ".bin")
        g = O.Gens(n, m)
        vals = [int.from_bytes(hashlib.shake_256(b"%s-v%d" % (name.encode(), i)).digest(8), "little") % (1 << n) for i in range(count * m)]
        if name.startswith("cfg1"):
            vals[0] = 1037578891   # README.md:120 of the reference
        bl = b"".join(hashlib.shake_256(b"%s-b%d" % (name.encode(), i)).digest(31) + b"\x00" for i in range(count * m))
        t0 = time.time()
        proofs, coms = O.prove_batch(g, vals, bl, m, n, LABEL, name.encode(), threads=threads)
        pl = O.proof_len(n, m)
        secs, verdicts, _ = O.verify_batch(g, proofs, coms, m, n, LABEL, hashlib.shake_256(b"chk").digest(64 * count), threads=threads)
        assert verdicts == bytes(count), "generated proofs must verify"
        with open(path, "wb") as f:
            f.write(b"BPBENCH1" + struct.pack("<IIIII", n, m, count, pl, len(LABEL)) + LABEL)
            for i in range(count):
                f.write(proofs[pl * i:pl * (i + 1)] + coms[32 * m * i:32 * m * (i + 1)])
        print("%s: %d proofs, %.1fs, %d bytes" % (name, count, time.time() - t0, os.path.getsize(path)))

if __name__ == "__main__":
    main()

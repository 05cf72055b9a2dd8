#!/usr/bin/env python3
"""Driver of tools/combine_rate.cpp: the pool's combining queue under plain host threads.
Writes the input file (512 proofs of (64, 1): even items proven on Transcript::new("combine-rate"), odd ones on transcripts that
already absorbed application messages -- three STROBE positions --, a few invalid; with the oracle's verdicts and advanced
transcripts), builds the client with g++ and runs the modes given on the command line:
    python tools/combine_rate.py [--seconds 3] [--lanes 8] [--window 0] "threads 64" "threads 1024" "tickets 16 64" "big 2 4096" ...
One JSON line per mode on stdout (rate, latency percentiles, proofs per chain, mismatches against the oracle)."""
import argparse
import hashlib
import os
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
OUT = os.path.join(ROOT, "gpurun_out")
INPUTS = os.path.join(ROOT, "bench_data", "combine_rate_inputs.bin")   # committed (written by make_inputs: `--make-inputs`)


def make_inputs(path, count=512, n=64, m=1):
    import pyoracle as O
    O.build()
    g = O.Gens(n, m)
    label = b"combine-rate"
    P, Cm, St, Rg, V, Ts, Fr = [], [], [], [], [], [], []
    for i in range(count):
        st = O.transcript_new(label)
        fresh = i % 2 == 0
        if not fresh:
            st = O.transcript_append_message(st, b"session", hashlib.shake_256(b"s%d" % i).digest(16))
            if i % 4 == 3:
                st = O.transcript_append_message(st, b"ctx", bytes([i & 0xff]) * 9)
            if i % 8 == 7:
                st, _ = O.transcript_challenge_bytes(st, b"binding", 16)
        pr, cm, _ = O.prove_ts(g, [int.from_bytes(hashlib.shake_256(b"v%d" % i).digest(8), "little")], hashlib.shake_256(b"b%d" % i).digest(31) + b"\0", n, st, b"cr%d" % i)
        pr = bytearray(pr)
        if i % 64 == 13:
            pr[129] ^= 2                    # wrong t_x
        if i % 128 == 77:
            pr[165:192] = b"\xff" * 27      # FormatError
        pr = bytes(pr)
        rng = hashlib.shake_256(b"r%d" % i).digest(64)
        rc, _, est = O.verify_ts(g, pr, cm, n, st, rng)
        P.append(pr), Cm.append(cm), St.append(st), Rg.append(rng), V.append(bytes([rc])), Ts.append(est if rc != 2 else st), Fr.append(bytes([1 if fresh else 0]))
    with open(path, "wb") as f:
        f.write(struct.pack("<4I", n, m, len(P[0]), count))
        for arr in (P, Cm, St, Rg, V, Ts, Fr):
            f.write(b"".join(arr))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--lanes", type=int, default=8)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--opts", default="")
    ap.add_argument("--make-inputs", action="store_true", help="(re)write bench_data/combine_rate_inputs.bin with the oracle and exit")
    ap.add_argument("modes", nargs="*", default=["threads 64", "threads 256", "threads 1024", "tickets 16 64", "big 2 4096"])
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if a.make_inputs:
        make_inputs(INPUTS)
        return 0
    inp = INPUTS
    if not os.path.exists(inp):
        inp = os.path.join(OUT, "combine_rate_inputs.bin")
        make_inputs(inp)
    exe = os.path.join(OUT, "combine_rate")
    lib = os.path.join(ROOT, "bulletproofs_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "combine_rate.cpp"),
                           "-L", lib, "-lbpgpu", "-Wl,-rpath," + lib, "-o", exe])
    env = dict(os.environ, BP_LANES=str(a.lanes))
    env.setdefault("GPU_MAX_HW_QUEUES", "16")
    if a.window:
        env["BP_W"] = str(a.window)
    if a.opts:
        env["BP_OPTS"] = a.opts
    rc_all = 0
    for mode in a.modes:
        r = subprocess.run([exe, inp, str(a.seconds)] + mode.split(), env=env)
        rc_all = rc_all or r.returncode
        sys.stdout.flush()
    return rc_all


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""bench.py -- 64-bit range-proof verifications/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
(`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under
torch.distributed.run with one rank per GPU.)

One "step" = one pass of the verification hot path (bpgpu_rangeproof_verify_batch_dev: proof bytes ->
Merlin transcript replay -> scalar expansion -> multiscalar multiplication -> verdict, all on the GPU)
over one batch of synthetic proofs per GPU.  Inputs are resident in HBM before the timed region; the
region is bracketed by barrier + synchronize; the maximum over ranks is reported.  Proofs are independent
units, so ranks share nothing during compute (weak scaling); the only collective is one all_gather of the
verdict bytes at the end (RCCL).

What keeps the number honest:
  * the fixture holds 8192 DISTINCT cfg2 proofs (256 / 512 distinct for cfg3 / cfg4); consecutive steps verify
    different 1024-slices, so the table lines a step gathers (~235 MB) are not the ones the previous steps left in
    the 256 MiB Infinity Cache;
  * every slice carries a few proofs with a flipped bit in t_x at fixed positions; after the timed region every
    verdict row must equal the expected 0/1 pattern of its slice -- a kernel that accepted (or rejected) everything
    fails the run.  No oracle is involved: the pattern follows from how the inputs were built;
  * when K steps cannot reach steady state (K < streams, or the region lasts < 0.25 s) the K-step region is
    repeated and the MEDIAN region time is reported (`regions` in the output); `steps` stays K.

Prints ONE JSON line on rank 0 (fields per the driver contract, plus `roofline` and `cpu_baseline`).
"""
import argparse
import hashlib
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The ROCm runtime maps HIP streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4); batches
# issued on different streams only overlap on the device when they sit on different hardware queues.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

L_ORDER = 2 ** 252 + 27742317777372353535851937790883648493


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3840)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4"],
                    help="BASELINE.json config; cfg2 (batch of 1024 single 64-bit proofs per GPU) is the metric's config")
    ap.add_argument("--batch", type=int, default=0, help="override proofs per GPU per step")
    ap.add_argument("--window-bits", type=int, default=0, help="fixed-base window (default: library default)")
    ap.add_argument("--table-bytes", type=int, default=0, help="HBM budget of the generator tables (default: library default)")
    ap.add_argument("--splits", type=int, default=0, help="workgroups the generator terms of a proof block are split over (default: library default)")
    ap.add_argument("--horner-lanes", type=int, default=0, choices=[0, 1, 4, 64], help="lanes per Horner chain (default: library default)")
    ap.add_argument("--streams", type=int, default=0,
                    help="(default 64) lanes of the library's pool (bpgpu_pool_create): independent (context, HIP stream) pairs its launch chains are "
                         "issued on round-robin, so that consecutive chains overlap on the device")
    ap.add_argument("--coalesce", type=int, default=0,
                    help="pool option coalesce_proofs: width the pool packs consecutive submitted batches into (default: the library's, 5120; "
                         "--coalesce = batch size means one launch chain per step, the round-2 behaviour)")
    ap.add_argument("--opt", default="", help="extra library options key=value[,key=value] (experiments, e.g. split_stage3=1)")
    ap.add_argument("--no-script", action="store_true", help="library option transcript_script = 0: byte-wise transcript replay (A/B of csrc/rp_script.h)")
    ap.add_argument("--direct", action="store_true",
                    help="bypass the pool: call bpgpu_rangeproof_verify_batch_dev on --streams (context, stream) pairs round-robin (the round-2 protocol, for A/B)")
    ap.add_argument("--bucket-min", type=int, default=0, help="bucket_min_terms option of the library (0 = default; a huge value forces the table-lookup path)")
    ap.add_argument("--single-process", action="store_true",
                    help="with --gpus N: ONE process, ONE pool over N devices (bpgpu_pool_create), host pointers in / verdicts out; N may exceed the "
                         "visible GPUs (shards then share devices: plumbing check)")
    ap.add_argument("--cfg5-only", type=int, default=0, help="run only the cfg5-shape MSM figure on this many streams and print it")
    ap.add_argument("--repeat", type=int, default=0, help="timed regions of K steps each (0 = auto: 1 when K steps reach steady state, else enough for ~1 s); the median is reported")
    ap.add_argument("--same-input", action="store_true", help="verify the SAME slice every step (the round-1 behaviour; for the cache A/B in DESIGN.md)")
    ap.add_argument("--rlc", action="store_true",
                    help="NOT the headline metric: time the batch-combination entry point bpgpu_rangeproof_verify_rlc_dev "
                         "(one combined identity check per batch, include/bpgpu.h) instead of the per-proof one")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the informational figures (batch-combined mode, cfg3, cfg5 shape)")
    ap.add_argument("--no-events", action="store_true", help="no kernel start/stop events in the timed region (no roofline block): measures their perturbation")
    ap.add_argument("--events-all", action="store_true", help="attach kernel start/stop events on every stream (default: every 32nd when steps >= 8 x streams)")
    return ap.parse_args()


def usable_cpus():
    """Logical CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(fx, threads=0):
    """The oracle (C restatement of the reference's algorithm: Straus below 190 terms / Pippenger above) timed on this box's host
    cores on a bounded sample of the same workload, one proof per thread, as many threads as the process may use (affinity and
    cgroup quota).  Rebuilt here with -march=native: on a CPU with AVX-512 IFMA that selects the 4-way vector backend of
    oracle/c/ifma4.h (parallel point formulas, the counterpart of the reference dependency's SIMD backends, README.md:69-84),
    otherwise the serial u64 5x51 one; the backend in use is named in `sample`.  This is the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    from bulletproofs_amd.workload import tile_batch
    native = O.build_native() if hasattr(O, "build_native") else None
    th = threads or usable_cpus()
    g = O.Gens(fx.n, fx.m)
    cal = max(4, min(64, fx.count))
    proofs, coms = tile_batch(fx, cal)
    rng = hashlib.shake_256(b"cpu-baseline").digest(64 * cal)
    t1, v, _ = O.verify_batch(g, proofs, coms, fx.m, fx.n, fx.label, rng, threads=1)   # single-thread calibration
    assert v == bytes(cal)
    per_proof = t1 / cal
    sample = int(max(th * 8, min(20.0 / per_proof, 65536)))   # ~20 s of CPU work
    proofs, coms = tile_batch(fx, sample)
    rng = hashlib.shake_256(b"cpu-baseline-%d" % th).digest(64 * sample)
    O.verify_batch(g, proofs[:fx.proof_len * th], coms[:32 * fx.m * th], fx.m, fx.n, fx.label, rng[:64 * th], threads=th)  # warm-up
    tN, v, _ = O.verify_batch(g, proofs, coms, fx.m, fx.n, fx.label, rng, threads=th)
    assert v == bytes(sample)
    return {"value": round(sample / tN, 1), "unit": "verifications/s", "cores": th, "kind": "port",
            "sample": "%d proofs (n=%d, m=%d) = %.1f s of CPU work on %d threads (%d logical CPUs visible, %d usable under the "
                      "cgroup quota); single thread: %.1f verifications/s (C restatement of the reference algorithm, "
                      "Straus<190<=Pippenger, field backend: %s; %s; not the Rust crate)"
                      % (sample, fx.n, fx.m, per_proof * sample, th, os.cpu_count() or 1, usable_cpus(), 1.0 / per_proof,
                         O.backend() if hasattr(O, "backend") else "u64 5x51 serial",
                         "built -march=native on this box" if native else "prebuilt -march=x86-64-v3"),
            "backend": O.backend() if hasattr(O, "backend") else "u64 5x51 serial"}


def plant_invalid(proofs, proof_len, batch, nslices):
    """Flip bit 0 of t_x (byte 128) of three proofs per slice at fixed, slice-dependent positions.  Returns the modified
    bytes and the expected verdict pattern [nslices][batch] (1 = VerificationError at the planted positions, else 0)."""
    pb = bytearray(proofs)
    expect = [bytearray(batch) for _ in range(nslices)]
    for j in range(nslices):
        for i in sorted({(37 * j + 11) % batch, (batch // 2 + 101 * j) % batch, batch - 1 - (j % batch)}):
            o = (j * batch + i) * proof_len + 128
            pb[o] ^= 1
            assert int.from_bytes(pb[o:o + 32], "little") < L_ORDER   # still a canonical scalar: VerificationError, not FormatError
            expect[j][i] = 1
    return bytes(pb), expect


class RangeProofBench:
    """One configuration on this rank's GPU: contexts/streams, inputs resident in HBM, timed regions."""

    def __init__(self, a, cfg, batch, nstreams, rank, local_dev, rlc=False):
        import torch
        import bulletproofs_amd as bp
        from bulletproofs_amd import workload as wl
        self.torch, self.bp, self.wl = torch, bp, wl
        self.a, self.cfg, self.batch, self.rlc, self.rank = a, cfg, batch, rlc, rank
        fx_name, _ = wl.CONFIGS[cfg]
        self.fx = fx = wl.load_fixture(fx_name)
        self.dev = dev = torch.device("cuda", local_dev)
        self.L = bp.lib()
        self.nslices = max(1, fx.count // batch)
        distinct = min(fx.count, self.nslices * batch)
        proofs, coms = wl.tile_batch(fx, self.nslices * batch)
        planted, self.expect_rows = plant_invalid(proofs, fx.proof_len, batch, self.nslices)
        to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        self.d_planted, self.d_clean, self.d_coms = to_dev(planted), to_dev(proofs), to_dev(coms)
        self.d_expect = torch.tensor([list(r) for r in self.expect_rows], dtype=torch.uint8, device=dev)
        self.d_rng = to_dev(hashlib.shake_256(b"bench-rng-%d" % rank).digest(64 * batch))
        self.d_wts = to_dev(hashlib.shake_256(b"bench-wts-%d" % rank).digest(64 * batch))
        self.distinct = distinct
        self.nstreams = nstreams
        self.ctxs, self.streams, self.pool = [], [], None
        self.use_pool = not a.direct   # (since round 4 the batch-combined check goes through the pool too: one combination per launch chain)
        if self.use_pool:
            # the library's scheduler: steps are SUBMITTED (device pointers) and the pool packs consecutive ones into launch chains
            self.pool = bp.Pool((local_dev,), nstreams, fixed_window_bits=a.window_bits or None, horner_lanes=a.horner_lanes or None,
                                fixed_splits=a.splits or None, fixed_table_max_bytes=a.table_bytes or None,
                                bucket_min_terms=a.bucket_min or None, coalesce_proofs=a.coalesce or None, transcript_script=0 if a.no_script else None)
            for kv in filter(None, a.opt.split(",")):
                self.pool.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            self.pool.gens_create(fx.n, fx.m)
        else:
            for _ in range(nstreams):
                c_ = bp.Context(local_dev, fixed_window_bits=a.window_bits or None, horner_lanes=a.horner_lanes or None,
                                fixed_splits=a.splits or None, fixed_table_max_bytes=a.table_bytes or None)
                if a.bucket_min:
                    c_.set_option("bucket_min_terms", a.bucket_min)
                if a.no_script:
                    c_.set_option("transcript_script", 0)
                c_.gens_create(fx.n, fx.m)
                self.ctxs.append(c_)
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        self.d_verdicts = None
        self.issued = 0

    def get_option(self, key):
        return self.pool.get_option(key) if self.pool else self.ctxs[0].get_option(key)

    def rlc_selfcheck(self):
        """batch-combined mode through the pool, outside the clock: a chain of two clean batches comes back all 0 with batch verdict 0; a
        chain holding one planted batch comes back undecided (5) for EVERY proof of the chain, batch verdict 1"""
        torch, fx = self.torch, self.fx
        v = torch.full((3, self.batch), 255, dtype=torch.uint8, device=self.dev)
        bo = torch.full((3, 36), 255, dtype=torch.uint8, device=self.dev)
        coms = self.d_coms.data_ptr()
        for r, src in ((0, self.d_clean), (1, self.d_clean)):
            self.pool.submit_rlc_dev(0, fx.n, fx.m, self.batch, src.data_ptr(), fx.proof_len, coms, fx.label, self.d_rng.data_ptr(), v[r].data_ptr(), bo[r].data_ptr())
        self.pool.wait()
        ok = bool((v[:2] == 0).all().item()) and int(bo[0][0].item()) == 0 and int(bo[1][0].item()) == 0
        self.pool.submit_rlc_dev(0, fx.n, fx.m, self.batch, self.d_clean.data_ptr(), fx.proof_len, coms, fx.label, self.d_rng.data_ptr(), v[0].data_ptr(), bo[0].data_ptr())
        self.pool.submit_rlc_dev(0, fx.n, fx.m, self.batch, self.d_planted.data_ptr(), fx.proof_len, coms, fx.label, self.d_rng.data_ptr(), v[2].data_ptr(), bo[2].data_ptr())
        self.pool.wait()
        ok = ok and bool((v[0] == 5).all().item()) and bool((v[2] == 5).all().item()) and int(bo[2][0].item()) == 1 and bool((bo[0][:33] == bo[2][:33]).all().item())
        if not ok:
            raise SystemExit("batch-combined check through the pool: self-check failed -- result invalid")

    def slice_of(self, g):
        """which slice of the fixture global step g verifies (ranks start at different slices)"""
        return 0 if self.a.same_input else (g + self.rank) % self.nslices

    def step(self, g, out_row, rlc=None):
        fx, L = self.fx, self.L
        rlc = self.rlc if rlc is None else rlc
        k = g % self.nstreams
        j = self.slice_of(g)
        # batch-combined mode: clean slices, except that every 8th step verifies a planted slice (expected: all undecided)
        planted = (not rlc) or (g % 8 == 7)
        base = (self.d_planted if planted else self.d_clean).data_ptr() + j * self.batch * fx.proof_len
        coms = self.d_coms.data_ptr() + j * self.batch * 32 * fx.m
        if self.pool is not None and not rlc:
            self.pool.submit_dev(0, fx.n, fx.m, self.batch, base, fx.proof_len, coms, fx.label, self.d_rng.data_ptr(), out_row.data_ptr())
            return
        if self.pool is not None:
            # batch-combined mode through the pool: clean slices only in the timed steps (a planted batch makes its whole CHAIN undecided,
            # which rlc_selfcheck verifies once, outside the clock)
            base = self.d_clean.data_ptr() + j * self.batch * fx.proof_len
            self.pool.submit_rlc_dev(0, fx.n, fx.m, self.batch, base, fx.proof_len, coms, fx.label, self.d_rng.data_ptr(), out_row.data_ptr())
            return
        if rlc:
            rc = L.bpgpu_rangeproof_verify_rlc_dev(self.ctxs[k].h, fx.n, fx.m, self.batch, base, fx.proof_len, coms, fx.label, len(fx.label),
                                                   self.d_rng.data_ptr(), self.d_wts.data_ptr(), out_row.data_ptr(), None, self.streams[k].cuda_stream)
        else:
            rc = L.bpgpu_rangeproof_verify_batch_dev(self.ctxs[k].h, fx.n, fx.m, self.batch, base, fx.proof_len, coms, fx.label, len(fx.label),
                                                     self.d_rng.data_ptr(), out_row.data_ptr(), None, self.streams[k].cuda_stream)
        if rc != 0:
            raise RuntimeError("bpgpu verify call failed: %s" % L.bpgpu_last_error(self.ctxs[k].h).decode())

    def expected(self, g0, K, rlc):
        """expected verdict rows of steps g0 .. g0+K-1 as a device tensor"""
        torch = self.torch
        idx = torch.tensor([self.slice_of(g0 + i) for i in range(K)], device=self.dev)
        e = self.d_expect[idx]
        if rlc and self.pool is not None:
            e = torch.zeros((K, self.batch), dtype=torch.uint8, device=self.dev)
        elif rlc:   # clean slices: all 0; planted ones: every proof undecided (5)
            planted = torch.tensor([1 if (g0 + i) % 8 == 7 else 0 for i in range(K)], dtype=torch.uint8, device=self.dev)
            e = (planted * 5).unsqueeze(1).expand(K, self.batch)
        return e

    def region(self, K, fence, gather=None, rlc=None):
        """issue K steps round-robin, wait; returns (seconds, host enqueue seconds).  Verdicts are checked against the
        planted pattern AFTER the clock stops."""
        torch = self.torch
        rlc = self.rlc if rlc is None else rlc
        if self.d_verdicts is None or self.d_verdicts.shape[0] < K:
            self.d_verdicts = torch.empty((max(K, 1), self.batch), dtype=torch.uint8, device=self.dev)
        self.d_verdicts.fill_(255)
        fence()                                              # the fill is complete before any other stream writes verdicts
        g0 = self.issued
        t0 = time.perf_counter()
        for i in range(K):
            self.step(g0 + i, self.d_verdicts[i], rlc)
        if self.pool is not None:
            self.pool.flush()                                # whatever the pool still holds back is issued now (inside the timed region)
        t_enq = time.perf_counter() - t0
        if gather is not None and self.pool is not None:
            self.pool.wait()                                 # the pool's streams are its own: wait for them before the collective reads the verdicts
        cur = torch.cuda.current_stream()
        for s_ in self.streams:
            cur.wait_stream(s_)                              # verdicts of every stream are complete before the gather
        allv = gather(self.d_verdicts[:K]) if gather else None
        fence()
        dt = time.perf_counter() - t0
        self.issued += K
        if K:
            exp = self.expected(g0, K, rlc)
            if not bool((self.d_verdicts[:K] == exp).all().item()):
                bad = int((self.d_verdicts[:K] != exp).sum().item())
                raise SystemExit("verdicts differ from the planted pattern in %d places -- result invalid" % bad)
        return dt, t_enq, allv

    def set_profile(self, on, every=1):
        if self.pool is not None:
            self.pool.profile_enable(on, every)
            return
        for c_ in self.ctxs:
            c_.profile_enable(False)
        if on:
            for c_ in self.ctxs[::every]:
                c_.profile_enable(True)

    def profile_reset(self):
        if self.pool is not None:
            self.pool.profile_reset()
        for c_ in self.ctxs:
            c_.profile_reset()

    def kernel_times(self):
        if self.pool is not None:
            return self.pool.profile_report()
        kern = {}
        for c_ in self.ctxs:
            for name, (cnt, ms) in c_.profile_report().items():
                o = kern.get(name, (0, 0.0))
                kern[name] = (o[0] + cnt, o[1] + ms)
        return kern

    def close(self):
        if self.pool is not None:
            self.pool.close()
            self.pool = None
        for c_ in self.ctxs:
            c_.close()
        self.ctxs = []


def timed(b, K, warmup, fence, repeat, gather=None, events_all=False, no_events=False, agree=None):
    """context set-up, warmup, then R regions of K steps; returns dict(elapsed (median), regions, enqueue, kern, allv)"""
    if b.pool is not None:                                   # context set-up (not a warmup step): the first chain on a lane sizes its arena
        b.region(max(K, 1), fence)                           # and caches the work decomposition -- one untimed region of the same shape
        if K >= 8 * b.nstreams:
            b.region(b.nstreams * 8, fence)
    else:
        for k in range(min(b.nstreams, max(K, 1))):
            b.step(k, _scratch_row(b))
    fence()
    if warmup:
        b.region(warmup, fence)
    b.profile_reset()
    if b.pool is not None:
        b.pool.set_option("stat_reset", 1)
    # start/stop events attached to a dispatch cost queue time: measured at the default workload, events on every 8th stream
    # lower the throughput by 4 % (5.25 vs 5.47 M/s, --no-events), so long runs sample every 32nd stream (~1 %)
    # (5.52 vs 5.50); a short burst pays more -- at --steps 20, events on all 20 streams cost 6 % (3.82 vs 4.05 M/s) -- so
    # short runs sample every 4th stream: with the >= 4 repeated regions that still gives launches >= steps
    sparse = (K >= 8 * b.nstreams) and not events_all
    every = 1 if events_all else max(1, min(32, b.nstreams // 4)) if sparse else max(1, min(4, b.nstreams // 4))
    if b.pool is not None and not events_all and not sparse:
        every = max(every, 8)      # a burst through the pool is ~5 chains on lanes 0..4: events on one of them (lane 0) per region
    b.set_profile(not no_events, every)
    regs, enq, allv = [], [], None
    dt, te, allv = b.region(K, fence, gather)
    regs.append(dt)
    enq.append(te)
    # the number of regions must be the same on every rank (each region ends in collectives): derive it from the slowest
    # rank's first region, not from the local clock
    dt_all = agree(dt) if agree else dt
    if repeat > 0:
        R = repeat
    elif K >= b.nstreams and dt_all >= 0.25:
        R = 1
    else:
        R = int(min(50, max(3, math.ceil(1.0 / max(dt_all, 1e-4)))))
    for _ in range(R - 1):
        dt, te, _ = b.region(K, fence, gather)
        regs.append(dt)
        enq.append(te)
    b.set_profile(False)
    return {"elapsed": statistics.median(regs), "regions": regs, "enqueue": statistics.median(enq), "kern": b.kernel_times(), "allv": allv,
            "events_every": every}


def _scratch_row(b):
    if b.d_verdicts is None:
        b.d_verdicts = b.torch.empty((1, b.batch), dtype=b.torch.uint8, device=b.dev)
    return b.d_verdicts[0]


# Measured ceilings of the box (profiles/r01_microbench_valu_rates.txt, tools/microbench.hip): the integer multiplier issues
# 3.035e13 v_mad_u64_u32 lane-operations/s chip-wide (5.18 cycles per wave-instruction per SIMD), and the engine's own mixed
# point addition (ge_madd, 1235 instructions, 707 of them v_mad_u64_u32) sustains 3.14e10/s at 4 waves per SIMD = 6.06e11
# wave-instructions/s of THIS instruction mix.  Both are reported; neither is a blend.
MAD_LANE_OPS_PER_S = 3.035e13
GE_MADD_WAVE_INSTR_PER_S = 3.14e10 * 1235 / 64
HBM_PEAK = 8.0e12
HBM_COPY_PEAK = 6.29e12       # measured device-to-device copy rate (MI355X_MICROARCH.md)


def _committed(name, cfg):
    path = os.path.join(ROOT, "profiles", "%s_%s.json" % (name, cfg))
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def roofline_block(cfg, n, m, kern, value, wl, events_every, proofs_per_launch, nsplit_note=""):
    """Roofline bookkeeping of one configuration.
    Top-level fields: the HBM roofline the contract asks for, of the kernel with the largest measured time -- ALGORITHMIC bytes per
    launch (32 N + 32 (4+2k+m) + 32 per verification, times the verifications one launch carries) over the launch's
    average duration, measured in this run by start/stop events attached to the dispatches.  The path is NOT HBM-bound -- ~10 field
    multiplications per input byte -- so the block also carries what binds (`valu`: two fractions of measured ceilings) and the
    counter-based HBM figures (`traffic`, `hbm_counter`: the window-table gathers are 50-70x the algorithmic bytes, on purpose).
    Fields marked "committed" are constants from the rocprofv3 --pmc passes under profiles/ (per verification), multiplied by this
    run's measured rate; everything else is measured live."""
    if not kern:
        return None
    ours = {k: v for k, v in kern.items() if k.startswith(("rp_", "finish", "rlc_", "fb_", "vb_", "bk_"))} or kern
    by_time = max(ours.items(), key=lambda kv: kv[1][1])[0]
    # The roofline kernel is the one that moves the most HBM bytes (committed counters) -- the table walk.  By summed (contended, overlapping)
    # kernel time another one can lead: since the Horner chains of wide launch chains run aside on a second stream, that is `rp_horner1`, 80
    # wavefronts per chain executing 350 k dependent instructions each -- 7 % of a chain's instructions, latency-bound by design.
    tr0 = _committed("pmc_traffic", cfg) or {}
    moved = {k: tr0[k] for k in ours if isinstance(tr0.get(k), (int, float))}
    dom = max(moved.items(), key=lambda kv: kv[1])[0] if moved else by_time
    cnt, ms = kern[dom]
    avg_s = ms / cnt * 1e-3
    alg_per_v = wl.algorithmic_bytes_per_verification(n, m)
    alg_bytes = alg_per_v * proofs_per_launch
    achieved = alg_bytes / avg_s / 1e9
    N = wl.msm_terms(n, m)
    out = {"bound": "valu",
           "bound_note": "what binds is 32-bit integer multiply issue (`valu`: two measured ceilings).  achieved / peak / unit / frac below are the "
                         "CONTRACT's HBM figures (algorithmic bytes per launch / live launch time against 8 TB/s) -- small by construction, ~10 field "
                         "multiplications per input byte; the counter-based HBM figures are `traffic` and `hbm_counter`",
           "hbm_contract_bound": "hbm",
           "kernel": dom, "dominant_by": ("largest HBM traffic per launch chain (committed FETCH_SIZE / WRITE_SIZE counters)" if moved else
                                          "largest total measured kernel time in this run"),
           "largest_summed_kernel_time": by_time,
           "achieved": round(achieved, 3), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved * 1e9 / HBM_PEAK,
           "algorithmic_bytes_per_launch": int(alg_bytes), "algorithmic_bytes_per_verification": alg_per_v,
           "verifications_per_launch": round(proofs_per_launch, 1), "avg_launch_us": round(avg_s * 1e6, 2), "launches": cnt,
           "timing": "start/stop events attached to the dispatches (hipExtLaunchKernelGGL) of every %slane of the pool, on their launch stream, inside the "
                     "timed region; kernel begin..end as in rocprofv3's kernel trace (durations are CONTENDED ones: other chains run beside)"
                     % ({1: "", 2: "2nd ", 3: "3rd "}.get(events_every, "%dth " % events_every)),
           "traffic": None}
    tr = _committed("pmc_traffic", cfg)
    if tr and tr.get("_proofs_per_launch"):
        per_v = {k: v / tr["_proofs_per_launch"] for k, v in tr.items() if not k.startswith("_") and k in kern}
        if dom in per_v:
            out["traffic"] = int(per_v[dom] * proofs_per_launch)
            out["traffic_note"] = ("HBM bytes of one %s launch from FETCH_SIZE / WRITE_SIZE (committed: profiles/pmc_traffic_%s.json, a rocprofv3 --pmc pass at %d "
                                   "verifications per launch, scaled to this run's launch width): %.0fx the algorithmic bytes -- the window-table lines that "
                                   "replace doublings" % (dom, cfg, tr["_proofs_per_launch"], per_v[dom] / alg_per_v))
            out["traffic_GBps_while_running"] = round(per_v[dom] * proofs_per_launch / avg_s / 1e9, 1)
        tot = sum(per_v.values())
        out["hbm_counter"] = {"bytes_per_verification_all_kernels": int(tot), "achieved_GBps": round(tot * value / 1e9, 1),
                              "frac_of_8TBps": tot * value / HBM_PEAK, "frac_of_measured_copy_peak": tot * value / HBM_COPY_PEAK,
                              "source": "committed counter bytes per verification x this run's measured rate"}
    wk = _committed("valu_work", cfg)
    if wk and wk.get("_proofs_per_launch"):
        wi = sum(v for k, v in wk.items() if not k.startswith("_") and k in kern) / wk["_proofs_per_launch"]
        fm = wk.get("_mad_u64_fraction", 0.58)
        out["valu"] = {"wave_instructions_per_verification": round(wi, 1), "mad_u64_fraction": fm,
                       "achieved_wave_instructions_per_s": wi * value,
                       "frac_of_mad_issue_peak": fm * wi * 64 * value / MAD_LANE_OPS_PER_S,
                       "frac_of_ge_madd_sustained": wi * value / GE_MADD_WAVE_INSTR_PER_S,
                       "peaks": {"v_mad_u64_u32_lane_ops_per_s": MAD_LANE_OPS_PER_S, "ge_madd_wave_instructions_per_s": round(GE_MADD_WAVE_INSTR_PER_S)},
                       "source": "SQ_INSTS_VALU per verification: committed (profiles/valu_work_%s.json, rocprofv3 --pmc pass); rate: measured in this run; "
                                 "peaks: measured microbenchmarks (profiles/r01_microbench_valu_rates.txt)" % cfg}
    out["point_ops"] = {"reference_count_per_verification": wl.reference_point_ops(N),
                        "reference_count_per_s": wl.reference_point_ops(N) * value,
                        "executed_per_verification": wl.executed_point_ops(n, m, nsplit_note[0], nsplit_note[1]) if nsplit_note else None,
                        "executed_per_s": wl.executed_point_ops(n, m, nsplit_note[0], nsplit_note[1]) * value if nsplit_note else None,
                        "note": "reference count = additions + doublings of the reference's own MSM algorithm for N = %d terms (SURVEY 8d: Straus below 190 terms, "
                                "Pippenger above); executed = what this engine performs per verification (window-table walk without doublings, 8-entry tables, "
                                "Horner chain, partial-sum reduction) -- reported beside, never instead" % N}
    out["kernels_us"] = {k: round(v[1] / v[0] * 1e3, 2) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])}
    return out


def mixed_shapes(a, local_dev, long_run):
    """ONE pool serving cfg3-shaped (m = 16) and cfg2-shaped (m = 1) proofs -- bpgpu_pool_gens_create(64, 16) +
    bpgpu_pool_gens_add_shape(64, 1): two window tables re-balanced under the one default budget.  Each shape's rate on the shared pool
    alone, then both in the same timed region (batches of both shapes submitted alternately); verdict rows checked against the planted
    pattern."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    dev = torch.device("cuda", local_dev)
    pool = bp.Pool((local_dev,), 64, fixed_table_max_bytes=a.table_bytes or None)
    pool.gens_create(64, 16)
    w_alone = pool.get_option("fixed_window_bits")
    pool.gens_add_shape(64, 1)
    out = {"windows": {"m16_alone": w_alone, "m16": pool.get_option("fixed_window_bits"), "m1": pool.get_option("secondary_window_bits")},
           "table_bytes": pool.get_option("fixed_table_bytes") + pool.get_option("secondary_table_bytes")}
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    sh = {}
    for key, cfg, batch in (("m1", "cfg2", 1024), ("m16", "cfg3", 256)):
        fx = wl.load_fixture(wl.CONFIGS[cfg][0])
        ns = max(1, fx.count // batch)
        proofs, coms = wl.tile_batch(fx, ns * batch)
        planted, rows = plant_invalid(proofs, fx.proof_len, batch, ns)
        sh[key] = dict(fx=fx, batch=batch, ns=ns, d_p=to_dev(planted), d_c=to_dev(coms), d_r=to_dev(hashlib.shake_256(b"mx" + key.encode()).digest(64 * batch)),
                       d_e=torch.tensor([list(r) for r in rows], dtype=torch.uint8, device=dev))

    def region(keys, K):
        bufs = {k: torch.full((K, sh[k]["batch"]), 255, dtype=torch.uint8, device=dev) for k in keys}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            for k in keys:
                x = sh[k]
                j = i % x["ns"]
                pool.submit_dev(0, x["fx"].n, x["fx"].m, x["batch"], x["d_p"].data_ptr() + j * x["batch"] * x["fx"].proof_len, x["fx"].proof_len,
                                x["d_c"].data_ptr() + j * x["batch"] * 32 * x["fx"].m, x["fx"].label, x["d_r"].data_ptr(), bufs[k][i].data_ptr())
        pool.wait()
        dt = time.perf_counter() - t0
        for k in keys:
            x = sh[k]
            exp = x["d_e"][torch.tensor([i % x["ns"] for i in range(K)], device=dev)]
            if not bool((bufs[k] == exp).all().item()):
                raise SystemExit("mixed shapes: verdicts differ from the planted pattern -- result invalid")
        return dt

    K = 256 if long_run else 24
    for keys in (("m1",), ("m16",), ("m1", "m16")):
        region(keys, K)                       # set-up (arenas, plans)
        dts = sorted(region(keys, K) for _ in range(3))
        name = "+".join(keys)
        out[name] = {k: round(sh[k]["batch"] * K / dts[1], 1) for k in keys}
    out["note"] = ("one pool, BulletproofGens(64, 16): m = 16 proofs walk the primary table, m = 1 proofs the secondary one; verifications/s of each shape "
                   "alone on the shared pool and of both submitted alternately in one timed region (both rates hold simultaneously there)")
    pool.close()
    return out


def _outliers(stderr_text):
    """tools/combine_rate.cpp's outlier report: every call above 5 x p99 with the time it returned at; `latency_ms.max` of all rows so far was
    a call of the first milliseconds (the first chain of a staging-buffer class allocates its pinned block and sizes a lane's arena)"""
    for ln in stderr_text.splitlines():
        if ln.startswith('{"outliers_above_5x_p99"'):
            try:
                o = json.loads(ln)
            except ValueError:
                return {}
            first = [k for k in o if k.startswith("of_them_in_first_")]
            return {"steady_latency_ms": o["steady_lat_ms"], "outliers_above_5x_p99": o["outliers_above_5x_p99"],
                    (first[0] if first else "of_them_at_start"): o[first[0]] if first else None,
                    "outliers_lat_ms_when_ms_thread_call": o["listed_as_lat_ms_when_ms_thread_call"][:8], "cgroup_cpu": o.get("cgroup_cpu")}
    return {}


def drop_in_call_shape(long_run, local_dev=0):
    """tools/combine_rate.cpp (built here with g++) against a pool of its own (W = 16 tables: 8.7 GB beside this process's): T host
    threads looping BLOCKING single-proof bpgpu_pool_rangeproof_verify_ts calls, tickets, and two threads with 4096-proof calls."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    inp = os.path.join(root, "bench_data", "combine_rate_inputs.bin")
    if not shutil.which("g++") or not os.path.exists(inp):
        return {"error": "g++ or bench_data/combine_rate_inputs.bin missing"}
    exe = os.path.join("/tmp", "bp_combine_rate_%d" % os.getpid())
    lib = os.path.join(root, "bulletproofs_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "combine_rate.cpp"),
                           "-L", lib, "-lbpgpu", "-Wl,-rpath," + lib, "-o", exe])
    env = dict(os.environ, BP_LANES="8", BP_W="16", GPU_MAX_HW_QUEUES="16")
    out = {"note": "plain host threads, one proof per blocking call, own `transcript: &mut Transcript` per proof (mod.rs:345-353); latency = call latency; "
                   "rate = threads / latency (Little): the device is far from full in this regime, see `tickets` and `two_callers_of_4096`"}
    secs = "2.0" if long_run else "1.0"
    for key, mode in (("threads_1", ["threads", "1"]), ("threads_64", ["threads", "64"]), ("threads_256", ["threads", "256"]),
                      ("tickets_16x128", ["tickets", "16", "128"]), ("two_callers_of_4096", ["big", "2", "4096"])):
        try:   # (the client has a watchdog of its own: a run that does not end prints the queue's state and exits)
            p = subprocess.run([exe, inp, secs] + mode, env=env, capture_output=True, text=True, timeout=90)
        except subprocess.TimeoutExpired:
            out[key] = {"error": "no result within 90 s"}
            continue
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if not line:
            out[key] = {"error": p.stderr[-1500:]}
            continue
        d = json.loads(line[-1])
        out[key] = {"verifications_per_s": d["rate_per_s"], "latency_ms": d["lat_ms"], "proofs_per_chain": d["proofs_per_chain"], "mismatches_vs_oracle": d["mismatches"],
                    "errors": d["errors"]}
        out[key].update(_outliers(p.stderr))
        if d["mismatches"] or d["errors"]:
            raise SystemExit("the combining queue returned a result that differs from the oracle's -- result invalid")
    # the boundary function itself, one multiscalar multiplication per blocking call (bpgpu_pool_msm_batch_shared; r1cs/verifier.rs:459-491's shape:
    # config 5, 4098 generator terms + 2081 points of the caller), every result compared with the committed oracle encodings
    try:
        sys.path.insert(0, os.path.join(root, "tools"))
        import make_msm_inputs
        mpath = os.path.join("/tmp", "bp_msm_inputs_%d.bin" % os.getpid())
        make_msm_inputs.write(mpath, local_dev)
        env_m = dict(env, BP_W="0", BP_MSM_INPUTS=mpath)   # (0 = the library's default: the largest table that fits its budget -- 146 GB, W = 15; this process holds no table of its own at this point)
        for key, mode in (("msm_threads_1", ["msm", "1", "1"]), ("msm_threads_64", ["msm", "64", "1"])):
            try:
                p = subprocess.run([exe, inp, secs] + mode, env=env_m, capture_output=True, text=True, timeout=90)
            except subprocess.TimeoutExpired:
                out[key] = {"error": "no result within 90 s"}
                continue
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if not line:
                out[key] = {"error": p.stderr[-1500:]}
                continue
            d = json.loads(line[-1])
            out[key] = {"msms_per_s": d["rate_per_s"], "latency_ms": d["lat_ms"], "msms_per_chain": d["proofs_per_chain"], "mismatches_vs_oracle": d["mismatches"],
                        "errors": d["errors"], "note": "6179-term MSMs, host pointers in and out (264 kB per MSM over PCIe), default tables for the 4098 generators (W = 15, 146 GB)"}
            out[key].update(_outliers(p.stderr))
            if d["mismatches"] or d["errors"]:
                raise SystemExit("a pooled multiscalar multiplication differs from the oracle's encoding -- result invalid")
        os.unlink(mpath)
    except SystemExit:
        raise
    except Exception as e:
        out["msm_threads_64"] = {"error": str(e)}
    try:
        os.unlink(exe)
    except OSError:
        pass
    return out


def bench_msm_small(local_dev):
    """The boundary function itself at the reference's own MSM size (optional_multiscalar_mul for ONE 64-bit single proof, N = 147;
    mod.rs:421): one blocking bpgpu_msm_batch call per MSM from one thread, host pointers in and out; points = party-0 generators of a (128, 1)
    set (valid encodings derived on the device), scalars = SHAKE256 mod l.  Latency only -- parity of this path is tests/test_gpu_msm.py's
    business (bit-exact vs the oracle incl. the golden sizes).  Informational (`extra`)."""
    import hashlib
    import bulletproofs_amd as bp
    L = 2**252 + 27742317777372353535851937790883648493
    hc = bp.Context(local_dev, fixed_window_bits=2)
    hc.gens_create(128, 1)
    G, H, _, _ = hc.gens_export()
    pts = (G + H)
    out = {"note": "one blocking bpgpu_msm_batch call, one MSM of N variable-base terms (no tables), p50 of 300 calls after 30; narrow form (msm_narrow, DESIGN 3.5) vs the batch form"}
    for n in (29, 147, 542):
        S = b"".join((int.from_bytes(hashlib.shake_256(b"msm-small-%d-%d" % (n, i)).digest(64), "little") % L).to_bytes(32, "little") for i in range(n))
        P = (pts * ((n + 255) // 256))[:32 * n]
        row = {}
        ref = None
        for form in (1, 0):
            hc.set_option("msm_narrow", form)
            for _ in range(30):
                res = hc.msm_batch([n], S, P)
            if ref is None:
                ref = res
            elif res != ref:
                raise SystemExit("bpgpu_msm_batch: the narrow and the batch form disagree -- result invalid")
            ts = []
            for _ in range(300):
                t0 = time.perf_counter()
                hc.msm_batch([n], S, P)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            row["ms_p50" if form else "ms_p50_batch_form"] = round(ts[150] * 1e3, 3)
        out["n_%d" % n] = row
    hc.close()
    return out


def bench_cfg5_shape(a, local_dev, steps=96, nstreams=16):
    """BASELINE config 5's MSM shape: N = 6179 = 4098 generator terms (tables) + 2081 per-MSM points, batches of 64 MSMs, inputs resident in
    HBM, through the library's pool: bpgpu_pool_msm_batch_shared_submit_dev issues every batch as one launch chain on the next of the pool's
    `nstreams` lanes (the (context, stream) pairs this function used to build by hand).  Informational (`extra`)."""
    import torch
    import bulletproofs_amd as bp
    dev = torch.device("cuda", local_dev)
    n, m, nb, nu = 2048, 1, 64, 2081
    ng = 2 * n * m + 2
    pool = bp.Pool((local_dev,), nstreams)
    if a.bucket_min:
        pool.set_option("bucket_min_terms", a.bucket_min)
    for kv in filter(None, a.opt.split(",")):
        pool.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    pool.gens_create(n, m)
    # inputs as SURVEY 8d specifies: uniform scalars; per-MSM points = RistrettoPoint::from_uniform_bytes outputs (the party-1 generator
    # chains of a (2048, 2) set, derived on the device by a small-table helper context) -- bulletproofs_amd/workload.py cfg5_inputs
    from bulletproofs_amd import workload as wl
    hc = bp.Context(local_dev, fixed_window_bits=2)
    hc.gens_create(n, 2)
    G2, H2, _, _ = hc.gens_export()
    hc.close()
    gsc, usc, upts = wl.cfg5_inputs(G2, H2, nb)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_gs, d_us, d_up = to_dev(gsc), to_dev(usc), to_dev(upts)
    d_out = torch.zeros((nstreams, nb, 32), dtype=torch.uint8, device=dev)
    d_st = torch.full((nstreams, nb), 255, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step(i, count=nb, out=None, ticket=False):
        k = i % nstreams
        return pool.msm_shared_submit_dev(0, n, m, count, nu, d_gs.data_ptr(), d_us.data_ptr(), d_up.data_ptr(), (out if out is not None else d_out[k]).data_ptr(),
                                          d_st[k].data_ptr(), want_ticket=ticket)
    for i in range(nstreams):
        step(i)
    pool.wait()
    pool.profile_reset()
    pool.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    pool.wait()
    dt = time.perf_counter() - t0
    # one lane alone: the latency of a single batch and of a single MSM
    t1 = time.perf_counter()
    step(0)
    pool.wait()
    one_batch = time.perf_counter() - t1
    d_out1 = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    step(0, 1, d_out1)
    pool.wait()
    pool.profile_enable(False)            # (no dispatch events on the latency measurement)
    kern = pool.profile_report()
    t2 = time.perf_counter()
    for q in range(10):
        step(q, 1, d_out1)                # (round-robin over ten lanes: what one caller with MSMs in a row sees)
    pool.wait()
    single_b2b = (time.perf_counter() - t2) / 10
    lat = []
    for _ in range(10):                   # one call at a time: enqueue + launch chain + wait
        t3 = time.perf_counter()
        step(0, 1, d_out1, True).wait()   # (the batch's own ticket: no walk over the other lanes)
        lat.append(time.perf_counter() - t3)
    single = sorted(lat)[5]
    ok = bool((d_st == 0).all().item()) and bool((d_out[0] == d_out[nstreams - 1]).all().item()) and bool((d_out[0] != 0).any().item())
    # MSM 0 and MSM 63 of the batch against the oracle's encodings, committed in bench_data/cfg5_expected.json (tools/gen_cfg5_expected.py)
    with open(os.path.join(ROOT, "bench_data", "cfg5_expected.json")) as f:
        exp5 = json.load(f)
    got = bytes(d_out[0].cpu().numpy().reshape(-1))
    ok = ok and got[:32].hex() == exp5["msm0"] and got[32 * (nb - 1):32 * nb].hex() == exp5["msm%d" % (nb - 1)] and bytes(d_out1[0].cpu().numpy()).hex() == exp5["msm0"]
    if not ok:
        raise SystemExit("cfg5-shape MSM: bad status, lanes disagree, or results differ from the committed oracle encodings -- result invalid")
    N = ng + nu
    alg = (32 * N + 32 * nu + 32) * nb
    by_time = max(kern.items(), key=lambda kv: kv[1][1])[0]
    dom, traffic = by_time, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_cfg5.json")
    if os.path.exists(tpath):      # the HBM roofline belongs to the kernel that moves the bytes (the table walk), not to the narrow one that leads by time
        with open(tpath) as f:
            tj = json.load(f)
        cand = {k: tj[k] for k in kern if isinstance(tj.get(k), (int, float))}
        if cand:
            dom = max(cand.items(), key=lambda kv: kv[1])[0]
            traffic = cand[dom]
    avg_s = kern[dom][1] / kern[dom][0] * 1e-3
    out = {"workload": "cfg5 shape: batches of %d MSMs of N = %d terms (%d generator terms from the tables + %d per-MSM from_uniform_bytes points, uniform scalars), %d streams; "
                       "MSM 0 and MSM %d checked against committed oracle encodings" % (nb, N, ng, nu, nstreams, nb - 1),
           "point_adds_per_s": None,
           "msms_per_s": round(nb * steps / dt, 1), "terms_per_s": round(nb * steps * N / dt, 1),
           "ms_per_batch_one_stream": round(one_batch * 1e3, 3), "ms_single_msm": round(single * 1e3, 3), "ms_single_msm_back_to_back": round(single_b2b * 1e3, 3),
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(alg / avg_s / 1e9, 3), "peak": 8000.0, "unit": "GB/s",
                        "frac": alg / avg_s / 1e9 / 8000.0, "traffic": traffic, "avg_launch_us": round(avg_s * 1e6, 2), "launches": kern[dom][0],
                        "algorithmic_bytes_per_launch": alg,
                        "kernels_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])}}}
    # what binds: instruction issue (committed SQ_INSTS_VALU of the chain's kernels at batch width x this run's rate), and the counter bytes of
    # the whole chain against the algorithmic ones
    wk = None
    wpath = os.path.join(ROOT, "profiles", "valu_work_cfg5.json")
    if os.path.exists(wpath):
        with open(wpath) as f:
            wk = json.load(f)
    if wk and wk.get("_proofs_per_launch"):
        wi = sum(v for k, v in wk.items() if not k.startswith("_") and k in kern and isinstance(v, (int, float))) / wk["_proofs_per_launch"]
        out["roofline"]["valu"] = {"wave_instructions_per_msm": round(wi, 1), "achieved_wave_instructions_per_s": wi * out["msms_per_s"],
                                   "frac_of_ge_madd_sustained": wi * out["msms_per_s"] / GE_MADD_WAVE_INSTR_PER_S,
                                   "source": "SQ_INSTS_VALU per MSM: committed (profiles/valu_work_cfg5.json, the batch-of-64 launches only); rate: measured in this run"}
    if os.path.exists(tpath) and tj.get("_proofs_per_launch"):
        tot = sum(v for k, v in tj.items() if not k.startswith("_") and k in kern and isinstance(v, (int, float))) / tj["_proofs_per_launch"]
        out["roofline"]["hbm_counter"] = {"bytes_per_msm_all_kernels": int(tot), "algorithmic_bytes_per_msm": alg // nb, "ratio": round(tot / (alg / nb), 1),
                                          "achieved_GBps": round(tot * out["msms_per_s"] / 1e9, 1), "frac_of_8TBps": tot * out["msms_per_s"] / HBM_PEAK,
                                          "note": "one 128-byte table line per generator-term addition is by design (33x of the ratio: 4098 terms x 17 windows x 128 B); "
                                                  "the sorted index lists of the bucket stage no longer leave LDS"}
    from bulletproofs_amd.workload import reference_point_ops
    out["point_adds_per_s"] = round(reference_point_ops(N) * out["msms_per_s"], 1)
    out["roofline"]["dominant_by"] = "most HBM bytes per launch (committed PMC pass); by total kernel time: %s" % by_time
    if traffic:
        out["roofline"]["traffic_GBps_while_running"] = round(traffic / avg_s / 1e9, 1)
    out["through"] = "bpgpu_pool_msm_batch_shared_submit_dev on a pool of %d lanes" % nstreams
    pool.close()
    return out


def self_launch(a):
    """`python bench.py --gpus N` outside torch.distributed.run: re-execute under it, one rank per GPU."""
    import socket
    with socket.socket() as sk:            # a port the kernel just handed out (a pid-derived one can collide on a shared box)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def lanes_for(a, steps, direct):
    """pool mode: --streams lanes (default 64).  Direct mode (round-2 protocol, batch-combined figures): 128 (context, stream) pairs,
    12 for a burst shorter than that (K = 20 on 5/8/10/12/16/20 streams: 3.3/3.8/4.25/4.27/4.22/4.05 M/s)."""
    if not direct:
        return a.streams if a.streams > 0 else 64
    if a.streams > 0:
        return max(1, min(a.streams, max(steps, 1)))
    return 128 if steps >= 128 else max(1, min(12, steps))


def single_process_multi_device(a):
    """--single-process with --gpus N: ONE process, ONE pool over N devices (bpgpu_pool_create(devices, N)), one host-pointer call per
    step for the whole N x batch proofs -- the C-ABI's own multi-device path (contiguous shard per device, host-side gather).
    Prints the same JSON line (PCIe-inclusive: this path hands over host buffers)."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx_name, default_batch = wl.CONFIGS[a.config]
    fx = wl.load_fixture(fx_name)
    batch = a.batch or default_batch
    ndev = torch.cuda.device_count()
    devices = [i % ndev for i in range(a.gpus)]
    pool = bp.Pool(devices, a.streams if a.streams > 0 else 32, fixed_window_bits=a.window_bits or None)
    pool.gens_create(fx.n, fx.m)
    total = batch * a.gpus * max(a.steps, 1)
    proofs, coms = wl.tile_batch(fx, total)
    planted, expect = plant_invalid(proofs, fx.proof_len, batch, total // batch)
    exp = b"".join(bytes(r) for r in expect)
    rng = hashlib.shake_256(b"sp-rng").digest(64 * total)
    for _ in range(max(1, min(a.warmup, 3))):
        pool.rangeproof_verify(fx.n, fx.m, planted, fx.proof_len, coms, fx.label, rng)
    regs = []
    for _ in range(a.repeat or 5):
        t0 = time.perf_counter()
        v = pool.rangeproof_verify(fx.n, fx.m, planted, fx.proof_len, coms, fx.label, rng)
        regs.append(time.perf_counter() - t0)
        if v != exp:
            raise SystemExit("verdicts differ from the planted pattern -- result invalid")
    dt = statistics.median(regs)
    print(json.dumps({"metric": "64-bit rangeproof verifications/sec (batched)", "value": round(total / dt, 1), "unit": "verifications/s", "n_gpus": a.gpus,
                      "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / max(a.steps, 1) * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "u32 limbs, u64 accumulators", "data": "synthetic (bench_data/%s.bin), host memory in, verdicts out" % fx_name,
                      "config": {"workload": "%s: ONE bpgpu_pool_rangeproof_verify call for %d proofs (%d per GPU per step x %d steps), single process over devices %s, "
                                             "host pointers (PCIe-inclusive)" % (a.config, total, batch, a.steps, devices), "mode": "single-process multi-device pool"},
                      "regions": [round(x, 5) for x in regs]}))
    pool.close()


def main():
    a = parse_args()
    if a.cfg5_only:
        print(json.dumps(bench_cfg5_shape(a, 0, 48, a.cfg5_only)))
        return
    if a.single_process:
        single_process_multi_device(a)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    import torch
    import torch.distributed as dist
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    from bulletproofs_amd import dist as bpdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    oversub = world > ndev            # more ranks than GPUs (plumbing check on a small box): RCCL refuses two ranks on one device -> gloo
    if world > 1:
        bpdist.init("gloo" if oversub else "nccl", None if oversub else dev)

    fx_name, default_batch = wl.CONFIGS[a.config]
    batch = a.batch or default_batch
    direct = a.direct
    nstreams = lanes_for(a, a.steps, direct)
    t_build0 = time.perf_counter()
    b = RangeProofBench(a, a.config, batch, nstreams, rank, local_dev, rlc=a.rlc)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build0          # contexts, generators derived on the device, window tables built (every rank its own)
    n, m = b.fx.n, b.fx.m
    N_terms = wl.msm_terms(n, m)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the final identity-check gather: one collective (RCCL; gloo on host copies when ranks share a GPU)
    gather = (lambda v: bpdist.gather_verdicts(v.cpu() if oversub else v, world)) if world > 1 else None
    agree = (lambda x: bpdist.max_over_ranks(x, world, None if oversub else dev)) if world > 1 else None
    r = timed(b, a.steps, a.warmup, fence, a.repeat, gather, a.events_all, a.no_events, agree)
    elapsed = bpdist.max_over_ranks(r["elapsed"], world, None if oversub else dev)
    if world > 1 and a.steps:     # every rank's verdict rows arrived and carry that rank's planted pattern
        allv = r["allv"]
        assert allv.shape[0] == world and bool(((allv == 0) | (allv == 1) | (allv == 5)).all().item())
    window_bits, table_bytes = b.get_option("fixed_window_bits"), b.get_option("fixed_table_bytes")
    value = world * batch * a.steps / elapsed if a.steps else 0.0
    # what a SCALE record needs to speak for itself: the ranks the process group really has, its backend, every rank's own
    # rate (its own clock, no barrier wait) and table-build time
    per_rank = bpdist.gather_floats([batch * a.steps / r["elapsed"] if a.steps and r["elapsed"] else 0.0, build_s, float(local_dev)], world, None if oversub else dev)
    multi = {"ranks_in_process_group": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
             "backend": (dist.get_backend() if (world > 1 and dist.is_initialized()) else "none (single process, no process group)"),
             "rccl_ranks": (dist.get_world_size() if (world > 1 and dist.is_initialized() and dist.get_backend() == "nccl") else 0),
             "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ or "LOCAL_RANK" in os.environ else "plain python",
             "visible_gpus": ndev, "ranks_share_gpus": oversub,
             "per_rank_verifications_per_s": [round(x[0], 1) for x in per_rank],
             "per_rank_table_build_s": [round(x[1], 3) for x in per_rank],
             "per_rank_device": [int(x[2]) for x in per_rank],
             "collective_in_timed_region": ("one all_gather of the verdict bytes per region + one MAX all-reduce of the region time" if world > 1 else "none"),
             "note": "value = n_gpus x batch x steps / max-over-ranks time; per-rank rates are each rank's own clock around the same regions.  N = 1 under "
                     "torch.distributed.run runs this same code without a process group (world = 1): its line equals the plain `python bench.py` line within box spread"}
    ppl, splits, sched = float(batch), 64, "one launch chain per step on %d (context, stream) pairs, round-robin (bpgpu_rangeproof_verify_batch_dev)" % nstreams
    if b.pool is not None:
        ch, cp = b.pool.get_option("stat_chains"), b.pool.get_option("stat_chain_proofs")
        ppl = cp / ch if ch else float(batch)
        splits = b.pool.get_option("stat_last_splits") or 64
        sched = ("library pool (bpgpu_pool_rangeproof_submit_dev + bpgpu_pool_flush): every step is submitted as its own batch with its own verdict buffer; "
                 "the pool packed consecutive steps into launch chains of %.0f proofs on average (coalesce_proofs = %d) over %d lanes"
                 % (ppl, b.pool.get_option("coalesce_proofs"), nstreams))
    roof = roofline_block(a.config, n, m, r["kern"], value / world, wl, r["events_every"], ppl, (window_bits, splits)) if rank == 0 else None
    b.close()

    extra = {}
    want_extra = world == 1 and not a.rlc and not a.no_extra and a.steps >= 8
    long_run = a.steps >= 640
    if want_extra:
        # (0) the same steps WITHOUT the pool's coalescing (round-2 protocol): one launch chain per step on (context, stream) pairs
        try:
            a.direct = True
            bd = RangeProofBench(a, a.config, batch, lanes_for(a, a.steps, True), rank, local_dev)
            rd = timed(bd, a.steps, a.warmup, fence, a.repeat, None, False, True)
            extra["direct_no_pool"] = {"verifications_per_s": round(batch * a.steps / rd["elapsed"], 1), "streams": bd.nstreams, "regions": len(rd["regions"]),
                                       "note": "bpgpu_rangeproof_verify_batch_dev, one launch chain per step (what round 2 reported as `value`)"}
            bd.close()
        except Exception as e:
            extra["direct_no_pool"] = {"error": str(e)}
        finally:
            a.direct = False
        # (1) the same batches through the batch-combined entry point (one identity check per batch) -- never `value`
        try:
            br = RangeProofBench(a, a.config, batch, lanes_for(a, a.steps, False), rank, local_dev, rlc=True)
            br.rlc_selfcheck()
            ks = max(64, a.steps // 4) if long_run else max(a.steps, 8)
            rr = timed(br, ks, 16 if long_run else 4, fence, 0, None, False, True)
            chr_, cpr = br.pool.get_option("stat_chains"), br.pool.get_option("stat_chain_proofs")
            extra["rlc"] = {"verifications_per_s": round(batch * ks / rr["elapsed"], 1), "steps": ks, "proofs_per_combination": round(cpr / chr_, 1) if chr_ else batch,
                            "note": "bpgpu_pool_rangeproof_submit_rlc_dev: batches of %d submitted to the pool, ONE combined identity check per launch chain "
                                    "(additional entry point, SURVEY 8f-3); a chain holding a planted batch comes back undecided as a whole (checked outside the clock)" % batch}
            br.close()
        except Exception as e:
            extra["rlc"] = {"error": str(e)}
    if want_extra and a.config == "cfg2" and not a.window_bits and not a.table_bytes:
        # the headline walks 113 GB of window tables (W = 20).  The same steps at W = 16 (8.7 GB, a budget a co-tenant can live with) and
        # W = 18 (33 GB): the curve the default is chosen from -- table bytes, build time, rate
        import copy
        curve = [{"fixed_window_bits": window_bits, "fixed_table_bytes": table_bytes, "build_s": round(build_s, 3), "verifications_per_s": round(value, 1),
                  "lookups_per_generator_term": -(-254 // window_bits) if window_bits else None}]
        for wb in (18, 16):
            try:
                aw = copy.copy(a)
                aw.window_bits = wb
                t_b0 = time.perf_counter()
                bw = RangeProofBench(aw, a.config, batch, lanes_for(a, a.steps, False), rank, local_dev)
                bs = time.perf_counter() - t_b0
                rw = timed(bw, a.steps, a.warmup, fence, a.repeat, None, False, True)
                row = {"fixed_window_bits": bw.get_option("fixed_window_bits"), "fixed_table_bytes": bw.get_option("fixed_table_bytes"), "build_s": round(bs, 3),
                       "verifications_per_s": round(batch * a.steps / rw["elapsed"], 1), "lookups_per_generator_term": -(-254 // wb), "regions": len(rw["regions"])}
                curve.append(row)
                if wb == 16:
                    extra["small_table"] = dict(row, note="the headline's steps with 16-bit windows: 16 table lookups per generator term instead of 13, a thirteenth of the HBM")
                bw.close()
            except Exception as e:
                curve.append({"fixed_window_bits": wb, "error": str(e)})
        extra["table_curve"] = {"rows": curve,
                                "note": "same steps, same box, the pool's window width as the only change: HBM spent on the generator tables against the rate (build_s: wall time of "
                                        "creating the pool, deriving the generators and building the table -- including the return of the previous pool's table to the allocator).  The "
                                        "default (largest table that fits the 160 GiB budget) buys the last ~10 % with 100 GB; a service that shares the device sets "
                                        "fixed_window_bits = 16 or 18 (or fixed_table_max_bytes) and keeps 90-97 % of the rate"}
    if want_extra and a.config == "cfg2" and not a.batch:
        # the batch-combined entry point at batches of 4096: from 32768 terms the per-proof points of the whole batch go through
        # ONE bucket (Pippenger) MSM (csrc/bucket.h)
        try:
            b4 = RangeProofBench(a, "cfg2", 4096, 32 if long_run else 12, rank, local_dev, rlc=True)
            k4 = 256 if long_run else max(a.steps, 8)
            r4 = timed(b4, k4, 32 if long_run else 8, fence, 0)
            extra["rlc_batch4096"] = {"verifications_per_s": round(4096 * k4 / r4["elapsed"], 1), "steps": k4, "streams": b4.nstreams, "regions": len(r4["regions"]),
                                      "kernels_us": {k_: round(v_[1] / v_[0] * 1e3, 2) for k_, v_ in sorted(r4["kern"].items(), key=lambda kv: -kv[1][1])},
                                      "note": "bpgpu_pool_rangeproof_submit_rlc_dev on batches of 4096 cfg2 proofs, one combination per batch, weights drawn by the library "
                                              "(69632 per-proof terms per combination: bucket MSM); `streams` = lanes of the pool"}
            b4.close()
        except Exception as e:
            extra["rlc_batch4096"] = {"error": str(e)}
    if want_extra and a.config == "cfg2":
        # (2) BASELINE configs 3 and 4 (aggregated m = 16 at batch 256, m = 32 at 512 per GPU) through the pool, each with its own roofline,
        # and (3) config 5's MSM shape
        for cfg_x, batch_x, m_x, terms_x in (("cfg3", 256, 16, 2090), ("cfg4", 512, 32, 4156)):
            try:
                bx = RangeProofBench(a, cfg_x, batch_x, lanes_for(a, a.steps, False), rank, local_dev)
                kx = (640 if cfg_x == "cfg3" else 256) if long_run else max(a.steps, 8)
                rx = timed(bx, kx, 64 if long_run else 8, fence, 0)
                vx = batch_x * kx / rx["elapsed"]
                chx, cpx = bx.pool.get_option("stat_chains"), bx.pool.get_option("stat_chain_proofs")
                wbx = bx.get_option("fixed_window_bits")
                extra[cfg_x] = {"workload": "%s: batches of %d aggregated m=%d 64-bit range proofs (MSM of %d terms each), %d distinct proofs, through the pool" % (cfg_x, batch_x, m_x, terms_x, bx.distinct),
                                "verifications_per_s": round(vx, 1), "point_adds_per_s": round(wl.reference_point_ops(terms_x) * vx, 1), "steps": kx, "regions": len(rx["regions"]),
                                "lanes": bx.nstreams, "fixed_window_bits": wbx,
                                "roofline": roofline_block(cfg_x, 64, m_x, rx["kern"], vx, wl, rx["events_every"], cpx / chx if chx else batch_x,
                                                           (wbx, bx.pool.get_option("stat_last_splits") or 64))}
                bx.close()
                if cfg_x == "cfg3" and long_run:
                    # the batch-combined check on the aggregated shape: one table MSM of 2050 terms per BATCH plus 40 points per proof.
                    # Its launches are narrow (256 proofs = 4 wavefronts of transcript lanes), so the rate follows the batch size
                    rl = {}
                    for b_r, st_r, k_r in ((256, 64, 640), (4096, 16, 64)):
                        br = RangeProofBench(a, "cfg3", b_r, st_r, rank, local_dev, rlc=True)
                        rr = timed(br, k_r, st_r, fence, 0, no_events=True)
                        rl["batch%d" % b_r] = round(b_r * k_r / rr["elapsed"], 1)
                        br.close()
                    extra[cfg_x]["rlc_verifications_per_s"] = rl
            except Exception as e:   # informational: never fails the headline line
                extra[cfg_x] = {"error": str(e)}
        try:
            extra["cfg5_shape"] = bench_cfg5_shape(a, local_dev)
        except SystemExit:
            raise
        except Exception as e:
            extra["cfg5_shape"] = {"error": str(e)}
        try:
            extra["msm_small_single_call"] = bench_msm_small(local_dev)
        except SystemExit:
            raise
        except Exception as e:
            extra["msm_small_single_call"] = {"error": str(e)}

    if want_extra and a.config == "cfg2" and not a.batch:
        # (4) the prover side: bpgpu_rangeproof_prove_batch, batches of 1024 single 64-bit proofs from host memory on
        # 4 host threads (one context each), every proof then verified by the engine -- variable time, see include/bpgpu.h
        try:
            import threading
            nbp, nthr = 1024, 4
            pvals = [int.from_bytes(hashlib.shake_256(b"pv%d" % i).digest(8), "little") for i in range(nbp)]
            pbl = hashlib.shake_256(b"pbl").digest(32 * nbp)
            prng = hashlib.shake_256(b"prng").digest(64 * (2 * 64 + 4) * nbp)
            pctx = []
            for _ in range(nthr):
                c_ = bp.Context(local_dev)
                c_.gens_create(64, 1)
                pctx.append(c_)
            res = [None] * nthr

            def work(k, reps):
                for _ in range(reps):
                    res[k] = pctx[k].rangeproof_prove_batch(64, 1, pvals, pbl, label=b"bench-prover", rng=prng)
            for reps in (1, 6):
                t1 = time.perf_counter()
                ths = [threading.Thread(target=work, args=(k, reps)) for k in range(nthr)]
                [t.start() for t in ths]
                [t.join() for t in ths]
                dtp = time.perf_counter() - t1
            okp = all(pctx[0].rangeproof_verify_batch(64, 1, r_[0], 672, r_[1], b"bench-prover") == bytes(nbp) for r_ in res)
            extra["prover"] = {"proofs_per_s": round(nthr * 6 * nbp / dtp, 1), "all_verify": okp,
                               "note": "bpgpu_rangeproof_prove_batch: %d host threads x 6 batches of %d single 64-bit proofs, host pointers, rng supplied; "
                                       "proofs byte-identical to the reference algorithm's (tests), variable time" % (nthr, nbp)}
            for c_ in pctx:
                c_.close()
            if not okp:
                raise SystemExit("prover output does not verify -- result invalid")
        except SystemExit:
            raise
        except Exception as e:
            extra["prover"] = {"error": str(e)}

    if want_extra and a.config == "cfg2" and not a.batch:
        # (5) the reference's literal call shape through the pool's combining queue (bpgpu_pool_rangeproof_verify_ts): plain host threads
        # (a C++ client, no Python in the loop), every proof with its own transcript (half of them pre-bound), every result compared with
        # the oracle's committed expectations (bench_data/combine_rate_inputs.bin).  PCIe-inclusive by nature; never `value`.
        try:
            extra["drop_in_call_shape"] = drop_in_call_shape(long_run, local_dev)
        except Exception as e:
            extra["drop_in_call_shape"] = {"error": str(e)}

    if want_extra and a.config == "cfg2" and not a.batch:
        try:
            extra["mixed_shapes"] = mixed_shapes(a, local_dev, long_run)
        except SystemExit:
            raise
        except Exception as e:
            extra["mixed_shapes"] = {"error": str(e)}

    if rank == 0:
        out = {
            "metric": "64-bit rangeproof verifications/sec (batched)" + (" -- batch-combined check (bpgpu_rangeproof_verify_rlc), not the headline mode" if a.rlc else ""),
            "value": round(value, 1),
            "unit": "verifications/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / max(a.steps, 1) * 1e3, 4),
            "point_adds_per_s": round(wl.reference_point_ops(N_terms) * value, 1),
            "point_adds_note": "the metric's second half, 'MSM point-adds/sec': A(N) x verifications/s with A(%d) = %d additions + doublings of the reference's own MSM "
                               "algorithm (SURVEY 8d); the engine's executed count is in roofline.point_ops" % (N_terms, wl.reference_point_ops(N_terms)),
            "regions": {"count": len(r["regions"]), "seconds": [round(x, 5) for x in r["regions"]],
                        "note": "timed regions of `steps` steps each; ms_per_step and value use the median region"},
            "host_enqueue_ms_per_step": round(r["enqueue"] / max(a.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (10x25.5-bit GF(2^255-19), 8x32-bit scalars mod l), u64 accumulators",
            "data": "synthetic (oracle-proved range proofs, bench_data/%s.bin: %d distinct proofs, a different %d-slice per step%s; 3 proofs per slice carry "
                    "a flipped bit and every verdict row is checked against that pattern)" % (fx_name, b.distinct, batch, " [--same-input: one slice]" if a.same_input else ""),
            "config": {"workload": "%s: batch of %d %s%d-bit range proofs per GPU per step, proof bytes -> verdict on device "
                                   "(MSM of %d terms each)" % (a.config, batch, ("aggregated m=%d " % m) if m > 1 else "single ", n, N_terms),
                       "n": n, "m": m, "batch_per_gpu": batch, "global_batch": batch * world, "msm_terms": N_terms,
                       "fixed_window_bits": window_bits, "fixed_table_bytes": table_bytes,
                       "mode": "rlc (one combined identity check per batch)" if a.rlc else "per-proof verdicts (the reference's semantics)",
                       "scheduler": sched, "lanes": nstreams, "parallelism": "independent proofs sharded, dp%d%s" % (world, " (ranks share GPUs: gloo gather)" if oversub else "")},
            "roofline": roof,
            "multi_gpu": multi,
        }
        if extra:
            out["extra"] = extra
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(b.fx, a.cpu_threads)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

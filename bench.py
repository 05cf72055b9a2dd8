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
    ap.add_argument("--horner-lanes", type=int, default=0, choices=[0, 4, 64], help="lanes per Horner chain (default: library default)")
    ap.add_argument("--streams", type=int, default=0,
                    help="(default 128; 12 for runs of fewer steps than that) independent (context, HIP stream) pairs the steps are issued on round-robin, so that "
                         "consecutive batches overlap on the device (one context per stream, as bpgpu.h prescribes "
                         "for concurrent callers)")
    ap.add_argument("--bucket-min", type=int, default=0, help="bucket_min_terms option of the library (0 = default; a huge value forces the table-lookup path)")
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
    """The oracle (C restatement of the reference's algorithm: u64 5x51 field, Straus/Pippenger split) timed on
    this box's host cores on a bounded sample of the same workload, one proof per thread, as many threads as the
    process may use (affinity and cgroup quota).  Built here with -march=native so the figure is not handicapped by
    the authoring container's ISA level.  This is the ONLY place bench.py touches oracle/."""
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
                      "cgroup quota); single thread: %.1f verifications/s (C restatement of the reference algorithm, u64 5x51 "
                      "field, Straus<190<=Pippenger, %s; not the Rust crate)"
                      % (sample, fx.n, fx.m, per_proof * sample, th, os.cpu_count() or 1, usable_cpus(), 1.0 / per_proof,
                         "built -march=native on this box" if native else "prebuilt -march=x86-64-v3")}


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
        self.ctxs = []
        for _ in range(nstreams):
            c_ = bp.Context(local_dev, fixed_window_bits=a.window_bits or None, horner_lanes=a.horner_lanes or None,
                            fixed_splits=a.splits or None, fixed_table_max_bytes=a.table_bytes or None)
            if a.bucket_min:
                c_.set_option("bucket_min_terms", a.bucket_min)
            c_.gens_create(fx.n, fx.m)
            self.ctxs.append(c_)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        self.d_verdicts = None
        self.issued = 0

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
        if rlc:   # clean slices: all 0; planted ones: every proof undecided (5)
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
        t_enq = time.perf_counter() - t0
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
        for c_ in self.ctxs:
            c_.profile_enable(False)
        if on:
            for c_ in self.ctxs[::every]:
                c_.profile_enable(True)

    def kernel_times(self):
        kern = {}
        for c_ in self.ctxs:
            for name, (cnt, ms) in c_.profile_report().items():
                o = kern.get(name, (0, 0.0))
                kern[name] = (o[0] + cnt, o[1] + ms)
        return kern

    def close(self):
        for c_ in self.ctxs:
            c_.close()
        self.ctxs = []


def timed(b, K, warmup, fence, repeat, gather=None, events_all=False, no_events=False, agree=None):
    """context set-up, warmup, then R regions of K steps; returns dict(elapsed (median), regions, enqueue, kern, allv)"""
    for k in range(min(b.nstreams, max(K, 1))):              # context set-up (not a warmup step): the first call on a context
        b.step(k, _scratch_row(b))                           # sizes its arena and caches the work decomposition
    fence()
    if warmup:
        b.region(warmup, fence)
    for c_ in b.ctxs:
        c_.profile_reset()
    # start/stop events attached to a dispatch cost queue time: measured at the default workload, events on every 8th stream
    # lower the throughput by 4 % (5.25 vs 5.47 M/s, --no-events), so long runs sample every 32nd stream (~1 %)
    # (5.52 vs 5.50); a short burst pays more -- at --steps 20, events on all 20 streams cost 6 % (3.82 vs 4.05 M/s) -- so
    # short runs sample every 4th stream: with the >= 4 repeated regions that still gives launches >= steps
    sparse = (K >= 8 * b.nstreams) and not events_all
    every = 1 if events_all else max(1, min(32, b.nstreams // 4)) if sparse else max(1, min(4, b.nstreams // 4))
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


VALU_ISSUE_CYCLES = {"mad_u64_u32": 5.18, "other": 4.0}   # measured, profiles/r01_microbench_valu_rates.txt (wave64 on a 16-lane SIMD: 4 cycles floor)


def roofline_block(cfg, n, m, batch, kern, value, wl, events_every, default_batch):
    """HBM roofline of the launch that carries the table walk (all of the path's HBM traffic worth the name and ~half of
    its VALU work), plus the VALU figure that actually binds.  Algorithmic bytes per verification at the MSM boundary
    (SURVEY.md 8d): 32 N + 32 (4+2k+m) + 32; one launch = `batch` of them."""
    if not kern:
        return None
    dom = "rp_stage4" if "rp_stage4" in kern else ("rlc_stage3" if "rlc_stage3" in kern else max(kern.items(), key=lambda kv: kv[1][1])[0])
    cnt, ms = kern[dom]
    avg_s = ms / cnt * 1e-3
    alg_bytes = wl.algorithmic_bytes_per_verification(n, m) * batch
    achieved = alg_bytes / avg_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % cfg)
    if os.path.exists(tpath) and batch == default_batch:   # HBM bytes/launch of this kernel from the committed rocprofv3 --pmc passes
        with open(tpath) as f:
            traffic = json.load(f).get(dom)
    valu = {}
    wpath = os.path.join(ROOT, "profiles", "valu_work_%s.json" % cfg)
    if os.path.exists(wpath) and batch == default_batch:
        with open(wpath) as f:
            wk = json.load(f)
        per_batch = sum(v for k, v in wk.items() if k in kern and not k.startswith("_"))
        frac_mad = wk.get("_mad_u64_fraction", 0.57)
        cyc = frac_mad * VALU_ISSUE_CYCLES["mad_u64_u32"] + (1 - frac_mad) * VALU_ISSUE_CYCLES["other"]
        peak = 1024 * 2.4e9 / cyc
        if per_batch:
            valu = {"wavefront_instructions_per_batch": per_batch,
                    "achieved_wavefront_instructions_per_s": per_batch * value / batch,
                    "peak_wavefront_instructions_per_s": peak,
                    "utilisation": per_batch * value / batch / peak,
                    "peak_note": "1024 SIMDs x 2.4 GHz / %.2f cycles per wave-instruction: the measured issue rates (v_mad_u64_u32 5.18 cycles, "
                                 "other VALU 4) weighted by the measured mix (%.0f %% v_mad_u64_u32, profiles/ pmc instruction mix)" % (cyc, 100 * frac_mad)}
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
            "frac": achieved / 8000.0, "traffic": traffic, "avg_launch_us": round(avg_s * 1e6, 2), "launches": cnt,
            "algorithmic_bytes_per_launch": alg_bytes,
            "timing": "start/stop events attached to the dispatches (hipExtLaunchKernelGGL) of every %sstream, on their launch stream, inside the "
                      "timed region; kernel begin..end as in rocprofv3's kernel trace" % ({1: "", 2: "2nd ", 3: "3rd "}.get(events_every, "%dth " % events_every)),
            "note": "the path is bound by integer VALU issue, not HBM: ~10 field multiplications per input byte, so the HBM fraction is ~1e-3 by construction",
            "dominant_by": "VALU work (half of a batch's wavefront-instructions) and all of the table traffic; by slot time under load the narrow, latency-bound "
                           "rp_stage1 (one lane per proof: 12 Keccak-f per proof) is comparable -- see kernels_us",
            "valu": valu,
            "kernels_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])}}


def bench_cfg5_shape(a, local_dev, steps=48, nstreams=8):
    """BASELINE config 5's MSM shape: N = 6179 = 4098 generator terms (tables) + 2081 per-MSM points, batches of 64 MSMs,
    through bpgpu_msm_batch_shared_dev; inputs resident in HBM.  Informational (`extra`)."""
    import torch
    import bulletproofs_amd as bp
    dev = torch.device("cuda", local_dev)
    L = bp.lib()
    n, m, nb, nu = 2048, 1, 64, 2081
    ng = 2 * n * m + 2
    ctxs = []
    for _ in range(nstreams):
        c_ = bp.Context(local_dev)
        if a.bucket_min:
            c_.set_option("bucket_min_terms", a.bucket_min)
        c_.gens_create(n, m)
        ctxs.append(c_)
    G, H, B, Bb = ctxs[0].gens_export()
    # scalars: uniform mod l (top nibble cleared keeps them canonical); points: the loaded generators in a scrambled order
    raw = bytearray(hashlib.shake_256(b"cfg5-scalars").digest(32 * (ng + nu) * nb))
    for i in range(31, len(raw), 32):
        raw[i] &= 0x0f
    gens = [G[32 * i:32 * i + 32] for i in range(n)] + [H[32 * i:32 * i + 32] for i in range(n)]
    upts = b"".join(gens[(7 * i + 3 * b) % len(gens)] for b in range(nb) for i in range(nu))
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_gs, d_us, d_up = to_dev(bytes(raw[:32 * ng * nb])), to_dev(bytes(raw[32 * ng * nb:])), to_dev(upts)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    d_out = torch.zeros((nstreams, nb, 32), dtype=torch.uint8, device=dev)
    d_st = torch.full((nstreams, nb), 255, dtype=torch.uint8, device=dev)

    def step(i):
        k = i % nstreams
        rc = L.bpgpu_msm_batch_shared_dev(ctxs[k].h, n, m, nb, nu, d_gs.data_ptr(), d_us.data_ptr(), d_up.data_ptr(), d_out[k].data_ptr(),
                                          d_st[k].data_ptr(), streams[k].cuda_stream)
        if rc != 0:
            raise RuntimeError("bpgpu_msm_batch_shared_dev failed: %s" % L.bpgpu_last_error(ctxs[k].h).decode())
    for i in range(nstreams):
        step(i)
    torch.cuda.synchronize()
    for c_ in ctxs:
        c_.profile_reset()
        c_.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # one stream alone: the latency of a single batch and of a single MSM
    t1 = time.perf_counter()
    step(0)
    torch.cuda.synchronize()
    one_batch = time.perf_counter() - t1
    d_out1 = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    L.bpgpu_msm_batch_shared_dev(ctxs[0].h, n, m, 1, nu, d_gs.data_ptr(), d_us.data_ptr(), d_up.data_ptr(), d_out1.data_ptr(), d_st[0].data_ptr(),
                                 streams[0].cuda_stream)
    torch.cuda.synchronize()
    for c_ in ctxs:
        c_.profile_enable(False)          # (no dispatch events on the latency measurement)
    t2 = time.perf_counter()
    for _ in range(10):
        L.bpgpu_msm_batch_shared_dev(ctxs[0].h, n, m, 1, nu, d_gs.data_ptr(), d_us.data_ptr(), d_up.data_ptr(), d_out1.data_ptr(), d_st[0].data_ptr(),
                                     streams[0].cuda_stream)
    torch.cuda.synchronize()
    single_b2b = (time.perf_counter() - t2) / 10
    lat = []
    for _ in range(10):                   # one call at a time: enqueue + launch chain + sync
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        L.bpgpu_msm_batch_shared_dev(ctxs[0].h, n, m, 1, nu, d_gs.data_ptr(), d_us.data_ptr(), d_up.data_ptr(), d_out1.data_ptr(), d_st[0].data_ptr(),
                                     streams[0].cuda_stream)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t3)
    single = sorted(lat)[5]
    ok = bool((d_st == 0).all().item()) and bool((d_out[0] == d_out[nstreams - 1]).all().item()) and bool((d_out[0] != 0).any().item())
    if not ok:
        raise SystemExit("cfg5-shape MSM: bad status or streams disagree -- result invalid")
    kern = {}
    for c_ in ctxs:
        c_.profile_enable(False)
        for name, (cnt, ms) in c_.profile_report().items():
            o = kern.get(name, (0, 0.0))
            kern[name] = (o[0] + cnt, o[1] + ms)
    N = ng + nu
    alg = (32 * N + 32 * nu + 32) * nb
    dom = max(kern.items(), key=lambda kv: kv[1][1])[0]
    avg_s = kern[dom][1] / kern[dom][0] * 1e-3
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_cfg5.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get(dom)
    out = {"workload": "cfg5 shape: batches of %d MSMs of N = %d terms (%d generator terms from the tables + %d per-MSM points), %d streams" % (nb, N, ng, nu, nstreams),
           "msms_per_s": round(nb * steps / dt, 1), "terms_per_s": round(nb * steps * N / dt, 1),
           "ms_per_batch_one_stream": round(one_batch * 1e3, 3), "ms_single_msm": round(single * 1e3, 3), "ms_single_msm_back_to_back": round(single_b2b * 1e3, 3),
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(alg / avg_s / 1e9, 3), "peak": 8000.0, "unit": "GB/s",
                        "frac": alg / avg_s / 1e9 / 8000.0, "traffic": traffic, "avg_launch_us": round(avg_s * 1e6, 2), "launches": kern[dom][0],
                        "algorithmic_bytes_per_launch": alg,
                        "kernels_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])}}}
    for c_ in ctxs:
        c_.close()
    return out


def self_launch(a):
    """`python bench.py --gpus N` outside torch.distributed.run: re-execute under it, one rank per GPU."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def main():
    a = parse_args()
    if a.cfg5_only:
        print(json.dumps(bench_cfg5_shape(a, 0, 48, a.cfg5_only)))
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
    auto_streams = a.streams <= 0
    if auto_streams:
        a.streams = 128
    nstreams = max(1, min(a.streams, max(a.steps, 1)))
    if a.steps < a.streams and auto_streams:
        # a burst shorter than the stream count never reaches the steady state the 128 streams are for: it is served best by
        # fewer streams than hardware queues (K = 20 on 5/8/10/12/16/20 streams: 3.3/3.8/4.25/4.27/4.22/4.05 M/s)
        nstreams = min(nstreams, 12)
    b = RangeProofBench(a, a.config, batch, nstreams, rank, local_dev, rlc=a.rlc)
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
    window_bits, table_bytes = b.ctxs[0].get_option("fixed_window_bits"), b.ctxs[0].get_option("fixed_table_bytes")
    value = world * batch * a.steps / elapsed if a.steps else 0.0

    extra = {}
    if world == 1 and not a.rlc and not a.no_extra and a.steps >= 8:
        # (1) the same batches through the batch-combined entry point (one identity check per batch) -- never `value`
        ks = max(b.nstreams, a.steps // 4)
        for k in range(b.nstreams):
            b.step(k, _scratch_row(b), True)
        fence()
        dt, _, _ = b.region(ks, fence, None, True)
        extra["rlc"] = {"verifications_per_s": round(batch * ks / dt, 1), "steps": ks,
                        "note": "bpgpu_rangeproof_verify_rlc_dev: one combined identity check per batch of %d (additional entry point, SURVEY 8f-3); "
                                "every 8th batch carries planted invalid proofs and must come back undecided" % batch}
    roof = roofline_block(a.config, n, m, batch, r["kern"], value, wl, r["events_every"], default_batch) if rank == 0 else None
    b.close()
    if world == 1 and not a.rlc and not a.no_extra and a.steps >= 8 and a.config == "cfg2" and not a.batch:
        # the batch-combined entry point at batches of 4096: from 32768 terms the per-proof points of the whole batch go through
        # ONE bucket (Pippenger) MSM (csrc/bucket.h)
        try:
            b4 = RangeProofBench(a, "cfg2", 4096, min(32, nstreams), rank, local_dev, rlc=True)
            k4 = 256 if a.steps >= 640 else max(a.steps, 8)
            r4 = timed(b4, k4, 32 if a.steps >= 640 else 8, fence, 0)
            extra["rlc_batch4096"] = {"verifications_per_s": round(4096 * k4 / r4["elapsed"], 1), "steps": k4, "streams": b4.nstreams, "regions": len(r4["regions"]),
                                      "kernels_us": {k_: round(v_[1] / v_[0] * 1e3, 2) for k_, v_ in sorted(r4["kern"].items(), key=lambda kv: -kv[1][1])},
                                      "note": "bpgpu_rangeproof_verify_rlc_dev on batches of 4096 cfg2 proofs (69632 per-proof terms per combination: bucket MSM)"}
            b4.close()
        except Exception as e:
            extra["rlc_batch4096"] = {"error": str(e)}
    if world == 1 and not a.rlc and not a.no_extra and a.steps >= 8 and a.config == "cfg2":
        # (2) BASELINE configs 3 and 4 (aggregated m = 16 at batch 256, m = 32 at 512 per GPU) and (3) config 5's MSM shape,
        # each with its own roofline
        for cfg_x, batch_x, streams_x, m_x, terms_x in (("cfg3", 256, 64, 16, 2090), ("cfg4", 512, 32, 32, 4156)):
            try:
                bx = RangeProofBench(a, cfg_x, batch_x, min(streams_x, nstreams), rank, local_dev)
                kx = (640 if cfg_x == "cfg3" else 256) if a.steps >= 640 else max(a.steps, 8)
                rx = timed(bx, kx, 64 if a.steps >= 640 else 8, fence, 0)
                vx = batch_x * kx / rx["elapsed"]
                extra[cfg_x] = {"workload": "%s: batch of %d aggregated m=%d 64-bit range proofs (MSM of %d terms each), %d distinct proofs" % (cfg_x, batch_x, m_x, terms_x, bx.distinct),
                                "verifications_per_s": round(vx, 1), "steps": kx, "regions": len(rx["regions"]), "streams": bx.nstreams,
                                "fixed_window_bits": bx.ctxs[0].get_option("fixed_window_bits"),
                                "roofline": roofline_block(cfg_x, 64, m_x, batch_x, rx["kern"], vx, wl, rx["events_every"], batch_x)}
                bx.close()
                if cfg_x == "cfg3" and a.steps >= 640:
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
        except Exception as e:
            extra["cfg5_shape"] = {"error": str(e)}

    if world == 1 and not a.rlc and not a.no_extra and a.steps >= 8 and a.config == "cfg2" and not a.batch:
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

    if world == 1 and not a.rlc and not a.no_extra and a.steps >= 8 and a.config == "cfg2" and not a.batch:
        # (5) the other public proof type on this path, LinearProof (src/linear_proof.rs): 4096 proofs of n = 64 made by
        # bpgpu_linear_create_batch over the context's generators, then verified (bases through the window tables); one planted
        # wrong commitment must be the only rejection
        try:
            nl, nbl = 64, 4096
            lc = bp.Context(local_dev)
            lc.gens_create(nl, 1)
            Gc, _, Bp_, Bb_ = lc.gens_export()
            ell = 2 ** 252 + 27742317777372353535851937790883648493
            sh_ = hashlib.shake_256(b"bench-linear").digest(32 * (2 * nl + 1) + 64)
            redl = lambda o: (int.from_bytes(sh_[o:o + 32] + bytes(32), "little") % ell).to_bytes(32, "little")
            la = b"".join(redl(32 * i) for i in range(nl))
            lb = b"".join(redl(32 * (nl + i)) for i in range(nl))
            lr = redl(64 * nl)
            lcc = sum(int.from_bytes(la[32 * i:32 * i + 32], "little") * int.from_bytes(lb[32 * i:32 * i + 32], "little") for i in range(nl)) % ell
            # C = <a, G> + r B + <a, b> F (linear_proof.rs:415-420) through the engine's own MSM
            Cl, stl = lc.msm_batch([nl + 2], la + lr + lcc.to_bytes(32, "little"), Gc[:32 * nl] + Bb_ + Bp_)
            assert stl == bytes(1)
            Cs_ = bytearray(Cl * nbl)
            Cs_[32 * 7:32 * 8] = Bp_                                  # proof 7 is checked against somebody else's commitment
            t1 = time.perf_counter()
            lproofs, lst = lc.linear_create_batch(nl, Cl * nbl, lr * nbl, la * nbl, lb, None, None, None, label=b"bench-linear")
            dtc = time.perf_counter() - t1
            pll = len(lproofs) // nbl
            lc.linear_verify_batch(nl, lproofs, pll, bytes(Cs_), None, None, None, lb, label=b"bench-linear")
            t1 = time.perf_counter()
            for _ in range(4):
                lv = lc.linear_verify_batch(nl, lproofs, pll, bytes(Cs_), None, None, None, lb, label=b"bench-linear")
            dtv = (time.perf_counter() - t1) / 4
            okl = lst == bytes(nbl) and [i for i in range(nbl) if lv[i]] == [7]
            extra["linear"] = {"verifications_per_s": round(nbl / dtv, 1), "proofs_created_per_s": round(nbl / dtc, 1), "verdicts_as_planted": okl,
                               "note": "LinearProof n = %d, batches of %d from host memory on ONE context: bpgpu_linear_create_batch (OS randomness, "
                                       "first call) then bpgpu_linear_verify_batch, both with the context's generators as bases (window tables)" % (nl, nbl)}
            lc.close()
            if not okl:
                raise SystemExit("linear-proof verdicts differ from the planted pattern -- result invalid")
        except SystemExit:
            raise
        except Exception as e:
            extra["linear"] = {"error": str(e)}

    if rank == 0:
        out = {
            "metric": "64-bit rangeproof verifications/sec (batched)" + (" -- batch-combined check (bpgpu_rangeproof_verify_rlc), not the headline mode" if a.rlc else ""),
            "value": round(value, 1),
            "unit": "verifications/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / max(a.steps, 1) * 1e3, 4),
            "regions": {"count": len(r["regions"]), "seconds": [round(x, 5) for x in r["regions"]],
                        "note": "timed regions of `steps` steps each; ms_per_step and value use the median region"},
            "host_enqueue_ms_per_step": round(r["enqueue"] / max(a.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (10x25.5-bit GF(2^255-19), 8x32-bit scalars mod l), u64 accumulators",
            "data": "synthetic (oracle-proved range proofs, bench_data/%s.bin: %d distinct proofs, a different %d-slice per step%s; 3 proofs per slice carry "
                    "a flipped bit and every verdict row is checked against that pattern)" % (fx_name, b.distinct, batch, " [--same-input: one slice]" if a.same_input else ""),
            "config": {"workload": "%s: batch of %d %s%d-bit range proofs per GPU, proof bytes -> verdict on device "
                                   "(MSM of %d terms each)" % (a.config, batch, ("aggregated m=%d " % m) if m > 1 else "single ", n, N_terms),
                       "n": n, "m": m, "batch_per_gpu": batch, "global_batch": batch * world, "msm_terms": N_terms,
                       "fixed_window_bits": window_bits, "fixed_table_bytes": table_bytes,
                       "mode": "rlc (one combined identity check per batch)" if a.rlc else "per-proof verdicts (the reference's semantics)",
                       "streams": nstreams, "parallelism": "independent proofs sharded, dp%d%s" % (world, " (ranks share GPUs: gloo gather)" if oversub else "")},
            "roofline": roof,
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

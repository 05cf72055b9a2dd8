#!/usr/bin/env python3
"""bench.py -- 64-bit range-proof verifications/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the verification hot path (bpgpu_rangeproof_verify_batch_dev: proof bytes ->
Merlin transcript replay -> scalar expansion -> multiscalar multiplication -> verdict, all on the GPU)
over one batch of synthetic proofs per GPU.  Inputs are resident in HBM before the timed region; the
region is bracketed by barrier + synchronize; the maximum over ranks is reported.  Proofs are independent
units, so ranks share nothing during compute (weak scaling); the only collective is one all_gather of the
verdict bytes at the end (RCCL).

Prints ONE JSON line on rank 0 (fields per the driver contract, plus `roofline` and `cpu_baseline`).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The ROCm runtime maps HIP streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4); batches
# issued on different streams only overlap on the device when they sit on different hardware queues.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3840)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4"],
                    help="BASELINE.json config; cfg2 (batch of 1024 single 64-bit proofs per GPU) is the metric's config")
    ap.add_argument("--batch", type=int, default=0, help="override proofs per GPU per step")
    ap.add_argument("--window-bits", type=int, default=0, help="fixed-base window (default: library default)")
    ap.add_argument("--splits", type=int, default=0, help="workgroups the generator terms of a proof block are split over (default: library default)")
    ap.add_argument("--horner-lanes", type=int, default=0, choices=[0, 4, 64], help="lanes per Horner chain (default: library default)")
    ap.add_argument("--streams", type=int, default=128,
                    help="independent (context, HIP stream) pairs the steps are issued on round-robin, so that "
                         "consecutive batches overlap on the device (one context per stream, as bpgpu.h prescribes "
                         "for concurrent callers)")
    ap.add_argument("--rlc", action="store_true",
                    help="NOT the headline metric: time the batch-combination entry point bpgpu_rangeproof_verify_rlc_dev "
                         "(one combined identity check per batch, include/bpgpu.h) instead of the per-proof one")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the informational batch-combined (rlc) figure")
    ap.add_argument("--events-outside", action="store_true",
                    help="collect the per-kernel HIP-event timings in a second pass instead of inside the timed region")
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


def cpu_baseline(fx, batch, threads=0):
    """The oracle (C restatement of the reference's algorithm: u64 5x51 field, Straus/Pippenger split) timed on
    this box's host cores on a bounded sample of the same workload, one proof per thread, as many threads as the
    process may use (affinity and cgroup quota).  This is the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    from bulletproofs_amd.workload import tile_batch
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
                      "field, Straus<190<=Pippenger; not the Rust crate)"
                      % (sample, fx.n, fx.m, per_proof * sample, th, os.cpu_count() or 1, usable_cpus(), 1.0 / per_proof)}


def main():
    a = parse_args()
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        bpdist.init("nccl", dev)

    fx_name, default_batch = wl.CONFIGS[a.config]
    fx = wl.load_fixture(fx_name)
    batch = a.batch or default_batch
    n, m = fx.n, fx.m
    N_terms = wl.msm_terms(n, m)

    nstreams = max(1, a.streams)
    ctxs = []
    for _ in range(nstreams):
        c_ = bp.Context(local_rank, fixed_window_bits=a.window_bits or None, horner_lanes=a.horner_lanes or None, fixed_splits=a.splits or None)
        c_.gens_create(n, m)
        ctxs.append(c_)
    ctx = ctxs[0]
    L = bp.lib()

    # this rank's shard of the global batch (weak scaling: `batch` proofs per GPU), resident in HBM
    lo = rank * batch
    proofs_b, coms_b = wl.tile_batch(fx, batch, first=lo)
    rng_b = hashlib.shake_256(b"bench-rng-%d" % rank).digest(64 * batch)
    d_proofs = torch.frombuffer(bytearray(proofs_b), dtype=torch.uint8).to(dev)
    d_coms = torch.frombuffer(bytearray(coms_b), dtype=torch.uint8).to(dev)
    d_rng = torch.frombuffer(bytearray(rng_b), dtype=torch.uint8).to(dev)
    d_verdicts = torch.full((max(a.steps, 1), batch), 255, dtype=torch.uint8, device=dev)
    d_wts = torch.frombuffer(bytearray(hashlib.shake_256(b"bench-wts-%d" % rank).digest(64 * batch)), dtype=torch.uint8).to(dev)
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(nstreams - 1)]

    def step(i, rlc=a.rlc):
        k = i % nstreams
        if rlc:
            rc = L.bpgpu_rangeproof_verify_rlc_dev(ctxs[k].h, n, m, batch, d_proofs.data_ptr(), fx.proof_len, d_coms.data_ptr(),
                                                   fx.label, len(fx.label), d_rng.data_ptr(), d_wts.data_ptr(),
                                                   d_verdicts[i % d_verdicts.shape[0]].data_ptr(), None, streams[k].cuda_stream)
        else:
            rc = L.bpgpu_rangeproof_verify_batch_dev(ctxs[k].h, n, m, batch, d_proofs.data_ptr(), fx.proof_len, d_coms.data_ptr(),
                                                     fx.label, len(fx.label), d_rng.data_ptr(),
                                                     d_verdicts[i % d_verdicts.shape[0]].data_ptr(), None, streams[k].cuda_stream)
        if rc != 0:
            raise RuntimeError("bpgpu_rangeproof_verify_batch_dev failed: %s" % L.bpgpu_last_error(ctxs[k].h).decode())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # context set-up (not a warmup step): the first call on a context sizes its arena and caches the work
    # decomposition, like gens_create above it allocates and synchronises
    for k in range(nstreams):
        step(k)
    fence()
    for i in range(a.warmup):
        step(i)
    fence()
    in_region_events = not a.events_outside
    # per-kernel timing inside the timed region samples every 8th (context, stream) pair: the start/stop events of a
    # dispatch are cheap but not free (~5 % of throughput when attached to every launch of every stream)
    prof_ctxs = ctxs[::8] if in_region_events else ctxs
    for c_ in ctxs:
        c_.profile_reset()
    for c_ in prof_ctxs:
        c_.profile_enable(in_region_events)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    t_enqueued = time.perf_counter() - t0
    for s_ in streams[1:]:
        streams[0].wait_stream(s_)                       # verdicts of every stream are complete before the gather
    all_v = bpdist.gather_verdicts(d_verdicts, world)   # the final identity-check gather: one collective
    fence()
    elapsed = time.perf_counter() - t0
    for c_ in ctxs:
        c_.profile_enable(False)
    elapsed = bpdist.max_over_ranks(elapsed, world, dev)
    ok = bool((all_v[:, :min(a.steps, all_v.shape[1])] == 0).all().item()) if a.steps else True
    if not ok:
        raise SystemExit("verification verdicts are not all Ok -- result invalid")

    if not in_region_events:   # second pass, same steps, only to time the kernels
        for c_ in ctxs:
            c_.profile_reset()
            c_.profile_enable(True)
        prof_ctxs = ctxs
        for i in range(a.steps):
            step(i)
        fence()
        for c_ in ctxs:
            c_.profile_enable(False)
    # informational second figure (single GPU, per-proof runs only): the same batches through the batch-combined entry
    # point bpgpu_rangeproof_verify_rlc_dev (one identity check per batch; include/bpgpu.h) -- never `value`
    extra = None
    if world == 1 and not a.rlc and not a.no_extra and a.steps >= 8:
        ksteps = max(nstreams, a.steps // 4)
        for k in range(nstreams):
            step(k, True)
        fence()
        d_verdicts.fill_(255)
        t1 = time.perf_counter()
        for i in range(ksteps):
            step(i, True)
        fence()
        dt = time.perf_counter() - t1
        if not bool((d_verdicts[:min(ksteps, d_verdicts.shape[0])] == 0).all().item()):
            raise SystemExit("batch-combined verdicts are not all Ok -- result invalid")
        extra = {"rlc_verifications_per_s": round(batch * ksteps / dt, 1), "rlc_steps": ksteps,
                 "note": "bpgpu_rangeproof_verify_rlc_dev: one combined identity check per batch of %d (additional entry point, "
                         "SURVEY 8f-3); not the headline mode" % batch}
    kern = {}
    for c_ in ctxs:
        for name, (cnt, ms) in c_.profile_report().items():
            o = kern.get(name, (0, 0.0))
            kern[name] = (o[0] + cnt, o[1] + ms)

    if rank == 0:
        value = world * batch * a.steps / elapsed
        # Dominant kernel and its roofline: the launch that carries the table walk (all of the path's HBM traffic
        # worth the name and ~half of its VALU work), else whatever took the most kernel time.  Algorithmic bytes
        # per verification at the MSM boundary: 32 N + 32 (4+2k+m) + 32; one launch = `batch` of them.
        dom = None
        if kern:
            dom = "rp_stage4" if "rp_stage4" in kern else ("rlc_stage3" if "rlc_stage3" in kern else max(kern.items(), key=lambda kv: kv[1][1])[0])
        roof = None
        if dom:
            cnt, ms = kern[dom]
            avg_s = ms / cnt * 1e-3
            alg_bytes = wl.algorithmic_bytes_per_verification(n, m) * batch
            achieved = alg_bytes / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % a.config)
            if os.path.exists(tpath):   # HBM bytes/launch of this kernel from the committed rocprofv3 --pmc passes
                with open(tpath) as f:
                    traffic = json.load(f).get(dom)
            # VALU utilisation: wavefront-instructions per batch (SQ_INSTS_VALU of the committed --pmc pass, summed over
            # the launches of one batch) x batches/s  /  (1024 SIMDs x one wave-instruction per 4 cycles at 2.4 GHz)
            valu_util = {}
            wpath = os.path.join(ROOT, "profiles", "valu_work_%s.json" % a.config)
            if os.path.exists(wpath) and batch == default_batch and not a.rlc:
                with open(wpath) as f:
                    wk = json.load(f)
                per_batch = sum(v for k, v in wk.items() if k in kern)
                if per_batch:
                    valu_util = {"wavefront_instructions_per_batch": per_batch,
                                 "achieved_wavefront_instructions_per_s": per_batch * value / batch,
                                 "peak_wavefront_instructions_per_s": 1024 * 2.4e9 / 4,
                                 "utilisation": per_batch * value / batch / (1024 * 2.4e9 / 4)}
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                    "frac": achieved / 8000.0, "traffic": traffic,
                    "avg_launch_us": round(avg_s * 1e6, 2), "launches": cnt,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "timing": "start/stop events attached to the dispatches (hipExtLaunchKernelGGL) of every 8th stream, on their launch stream, %s the timed region; "
                              "kernel begin..end as in rocprofv3's kernel trace" % ("inside" if in_region_events else "second pass after"),
                    # the binding resource is integer VALU issue, not HBM: also report it
                    "valu": {**valu_util,
                             "reference_point_ops_per_s": wl.reference_point_ops(N_terms) * value,
                             "measured_peak_madd_per_s": 3.14e10,
                             "note": "A(N)=%d point ops per MSM by the reference's own algorithm x verifications/s; "
                                     "peak = ge_madd microbenchmark (profiles/r01_microbench_valu_rates.txt)" % wl.reference_point_ops(N_terms)},
                    "kernels_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])}}
        out = {
            "metric": "64-bit rangeproof verifications/sec (batched)" + (" -- batch-combined check (bpgpu_rangeproof_verify_rlc), not the headline mode" if a.rlc else ""),
            "value": round(value, 1),
            "unit": "verifications/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / max(a.steps, 1) * 1e3, 4),
            "host_enqueue_ms_per_step": round(t_enqueued / max(a.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (10x25.5-bit GF(2^255-19), 8x32-bit scalars mod l), u64 accumulators",
            "data": "synthetic (oracle-proved range proofs, bench_data/%s.bin, tiled to the batch; all verdicts checked Ok)" % fx_name,
            "config": {"workload": "%s: batch of %d %s%d-bit range proofs per GPU, proof bytes -> verdict on device "
                                   "(MSM of %d terms each)" % (a.config, batch, ("aggregated m=%d " % m) if m > 1 else "single ", n, N_terms),
                       "n": n, "m": m, "batch_per_gpu": batch, "global_batch": batch * world, "msm_terms": N_terms,
                       "fixed_window_bits": ctx.get_option("fixed_window_bits"), "fixed_table_bytes": ctx.get_option("fixed_table_bytes"),
                       "mode": "rlc (one combined identity check per batch)" if a.rlc else "per-proof verdicts (the reference's semantics)",
                       "streams": nstreams, "parallelism": "independent proofs sharded, dp%d" % world},
            "roofline": roof,
        }
        if extra:
            out["extra"] = extra
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fx, batch, a.cpu_threads)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c_ in ctxs:
        c_.close()


if __name__ == "__main__":
    main()

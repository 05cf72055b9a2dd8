#!/usr/bin/env python3
"""8-GPU dry run on the ONE GPU of the box, under the box's real CPU quota: the two multi-device forms with
eight shards / eight ranks all mapped onto device 0 (tables at W = 16 so that eight of them fit), with the host CPU seconds each
form burns -- the question being whether 8 pools' worth of host threads starve the caller under a 16-CPU cgroup -- and the
N = 1 torchrun form next to the plain N = 1 bench (must agree within box spread).
    python tools/dryrun8.py   ->  JSON lines on stdout"""
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, GPU_MAX_HW_QUEUES="16", HSA_ENABLE_IPC_MODE_LEGACY="0")


def run(name, cmd, timeout=900):
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True, env=ENV, timeout=timeout, cwd=ROOT)
    dt = time.time() - t0
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            line = json.loads(ln)
    out = {"form": name, "rc": p.returncode, "wall_s": round(dt, 1), "host_cpu_user_s": round(r1.ru_utime - r0.ru_utime, 1),
           "host_cpu_sys_s": round(r1.ru_stime - r0.ru_stime, 1), "cpus_granted": len(os.sched_getaffinity(0))}
    if line:
        out.update({k: line.get(k) for k in ("value", "n_gpus", "steps", "ms_per_step")})
        out["workload"] = (line.get("config") or {}).get("workload", "")[:160]
    else:
        out["stderr_tail"] = p.stderr[-600:]
    print(json.dumps(out), flush=True)


def main():
    py = sys.executable
    b = [py, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--window-bits", "16"]
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        print(json.dumps({"cgroup_cpu_max": q, "nproc": os.cpu_count()}), flush=True)
    except OSError:
        pass
    run("N=1 plain", b + ["--steps", "20", "--warmup", "5"])
    run("N=1 torchrun", [py, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29711",
                         os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--window-bits", "16", "--gpus", "1", "--steps", "20", "--warmup", "5"])
    run("8 ranks on device 0 (torchrun form, gloo gather)", b + ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    run("8 shards on device 0 (single process, one pool, one host call per step)", b + ["--gpus", "8", "--single-process", "--steps", "8", "--streams", "8"])
    run("2 shards on device 0 (single process)", b + ["--gpus", "2", "--single-process", "--steps", "8", "--streams", "16"])


if __name__ == "__main__":
    main()

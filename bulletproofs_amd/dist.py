"""Multi-GPU plumbing: one process per GPU, proofs sharded by contiguous ranges, no data-path
collective; the only exchange is the final gather of verdict bytes (RCCL when the tensors are on
GPUs, gloo in the CPU tests).  SURVEY.md section 8(e)."""
import os


def init(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend == "gloo":   # single-node use: bind the loopback device instead of resolving the (possibly unresolvable) hostname
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    if backend == "nccl" and device is not None:
        dist.init_process_group(backend, device_id=device)
    else:
        dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def gather_verdicts(local_verdicts, world):
    """all_gather of equal-shaped uint8 verdict tensors; returns a [world, ...] tensor on every rank."""
    import torch
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_verdicts.unsqueeze(0)
    out = [torch.empty_like(local_verdicts) for _ in range(world)]
    dist.all_gather(out, local_verdicts)
    return torch.stack(out)


def max_over_ranks(value, world, device=None):
    import torch
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(values, world, device=None):
    """every rank's list of floats (equal lengths) -> [world][len] on every rank: the SCALE record's per-rank figures (rate, table build time)"""
    import torch
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return [[float(v) for v in values]]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [[float(x) for x in o.cpu().tolist()] for o in out]

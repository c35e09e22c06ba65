"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU).

`hifimeth call` shards by READ: every read is independent (reference mod_main.cpp:180-212), so a
rank owns whole read slabs and the data path needs no collective.  The only exchanges are the job
throughput reduction (sum of sites, max of time) and, when one rank writes the output, the gather
of the per-rank call records in slab order.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def gpu_count_no_init() -> int:
    """GPUs this process would see, counted WITHOUT opening the device (a launcher or a barrier-only rank must not become one more
    process holding the GPU): KFD topology nodes with SIMDs, cut down by *_VISIBLE_DEVICES.  -1 = cannot tell (no KFD sysfs)."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return -1

    for path in nodes:
        try:
            for line in open(path):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            return -1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def gpu_count() -> int:
    """gpu_count_no_init(), or torch's count where the KFD topology cannot be read."""
    n = gpu_count_no_init()
    if n < 0:
        import torch
        n = torch.cuda.device_count()
    return n


def init_process_group(backend: str | None = None, force: bool = False):
    """Initialise torch.distributed from the env (MASTER_ADDR/PORT, RANK, WORLD_SIZE). Returns the
    module, or None for a single-process run (`force`: build a world of one anyway, to rehearse the collectives)."""
    rank, local_rank, world = env_world()
    if world <= 1 and not force:
        return None
    if world <= 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        # (this pool's host driver supports dmabuf IPC only; without the setting RCCL fails with `hipIpcGetMemHandle: invalid argument`.
        #  It must be in the environment before the first HIP call of the process: the launchers export it, this is the last chance)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def slab_assignment(n_slabs: int, rank: int, world: int) -> List[int]:
    """Static interleave of read slabs over ranks (slabs have similar size, so this balances)."""
    return list(range(rank, n_slabs, world))


def make_slabs(n_reads: int, reads_per_slab: int) -> List[range]:
    return [range(s, min(n_reads, s + reads_per_slab)) for s in range(0, n_reads, reads_per_slab)]


def job_throughput(dist, sites_local: float, seconds_local: float, device: str = "cpu"):
    """Whole-job (sites, seconds): SUM of sites over ranks, MAX of elapsed time over ranks."""
    if dist is None:
        return float(sites_local), float(seconds_local)
    import torch
    s = torch.tensor([float(sites_local)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(s.item()), float(t.item())


def gather_rank_stats(dist, values: Sequence[float], device: str = "cpu") -> List[List[float]]:
    """Every rank's row of floats, on every rank (all_gather): [rank][value]."""
    if dist is None:
        return [[float(v) for v in values]]
    import torch
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, mine)
    return [[float(x) for x in r.cpu()] for r in rows]


def gather_calls(dist, per_slab_calls: Sequence[np.ndarray], slab_ids: Sequence[int], n_slabs: int, dtype):
    """Collect the call records of every slab on rank 0, concatenated in slab (= input read) order,
    the order the reference writes its output in (mod_main.cpp:353-354). Other ranks get None."""
    if dist is None:
        order = np.argsort(np.asarray(slab_ids)) if len(slab_ids) else []
        parts = [per_slab_calls[i] for i in order]
        return np.concatenate(parts) if parts else np.empty(0, dtype)
    payload = [(int(sid), np.asarray(c).tobytes()) for sid, c in zip(slab_ids, per_slab_calls)]
    gathered = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(payload, gathered, dst=0)
    if dist.get_rank() != 0:
        return None
    by_slab = {}
    for part in gathered:
        for sid, raw in part:
            by_slab[sid] = np.frombuffer(raw, dtype=dtype)
    assert sorted(by_slab) == list(range(n_slabs)), "a slab is missing or duplicated"
    parts = [by_slab[s] for s in range(n_slabs)]
    return np.concatenate(parts) if parts else np.empty(0, dtype)

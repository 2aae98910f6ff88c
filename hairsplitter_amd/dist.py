"""Multi-GPU plumbing of the hot path: contigs are independent units, so a batch is sharded by
contig over the ranks (one process per GPU). Two small exchanges exist: the global error rate that stage 4 needs
(mean of the per-contig generate_msa distances, call_variants.cpp:1310-1316,1377) and the final gather of the
partition labels to rank 0, which writes the .gro. `torch.distributed` backend "nccl" is RCCL on ROCm; the
same code runs on "gloo" for the CPU tests."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def lpt_shards(weights: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time assignment of contigs (weight = aligned bp) to ranks; deterministic."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    load = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += weights[i]
    for s in shards:
        s.sort()
    return shards


def global_error_rate(local_ids: Sequence[int], local_mean_distance: np.ndarray, n_contigs_total: int, group=None) -> float:
    """Sum of the per-contig mean distances (> 0 only) in *contig index order* with float32 accumulation, divided by
    the number of such contigs -- what a 1-thread reference run prints to error_rate.txt."""
    import torch
    import torch.distributed as dist
    full = torch.zeros(n_contigs_total, dtype=torch.float32)
    if len(local_ids):
        full[torch.as_tensor(list(local_ids), dtype=torch.long)] = torch.from_numpy(np.asarray(local_mean_distance, np.float32))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        full = full.to(dev)
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)   # every contig is owned by exactly one rank: exact
        full = full.cpu()
    total = np.float32(0.0)
    n = 0
    for v in full.numpy():
        if v > 0:
            total = np.float32(total + v)
            n += 1
    return float(np.float32(total / np.float32(n))) if n else float("nan")


def gather_labels(labels: np.ndarray, group=None, dst: int = 0):
    """The single gather of partition labels at the end (int16 on the wire: labels are -2, -1 or a group id < N).
    Returns the list of per-rank label arrays on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [labels]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    assert labels.size == 0 or (labels.min() >= -2 and labels.max() < 32767)
    n = torch.tensor([labels.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    mx = int(max(int(s.item()) for s in sizes))
    wire = torch.int16 if dist.get_backend(group) == "nccl" else torch.int32   # gloo has no int16 gather
    buf = torch.full((max(mx, 1),), -2, dtype=wire, device=dev)
    if labels.size:
        buf[:labels.size] = torch.from_numpy(labels.astype(np.int32)).to(dev).to(wire)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return None
    return [o[:int(s.item())].cpu().numpy().astype(np.int32) for o, s in zip(out, sizes)]

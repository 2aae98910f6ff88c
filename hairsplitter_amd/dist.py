"""Multi-GPU plumbing of the hot path: contigs are independent units, so a batch is sharded by
contig over the ranks (one process per GPU). Two small exchanges exist: the global error rate that stage 4 needs
(mean of the per-contig generate_msa distances, call_variants.cpp:1310-1316,1377) and the final gather of the
partition labels to rank 0, which writes the .gro. `torch.distributed` backend "nccl" is RCCL on ROCm; the
same code runs on "gloo" for the CPU tests."""
from __future__ import annotations

from typing import Optional, List, Sequence

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
    # `group` may be a gloo group on the same ranks: the payload is a few hundred floats, a host-side exchange is enough
    """Sum of the per-contig mean distances (> 0 only) in *contig index order* with float32 accumulation, divided by
    the number of such contigs -- what a 1-thread reference run prints to error_rate.txt."""
    import torch
    import torch.distributed as dist
    full = torch.zeros(n_contigs_total, dtype=torch.float32)
    if len(local_ids):
        full[torch.as_tensor(list(local_ids), dtype=torch.long)] = torch.from_numpy(np.asarray(local_mean_distance, np.float32))
    if dist.is_available() and dist.is_initialized():
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        full = full.to(dev)
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)   # every contig is owned by exactly one rank: exact
        full = full.cpu()
    return mean_of_positive_f32(full.numpy())


def global_window_size(local_rec_refspan: np.ndarray, amplicon: bool = False, longest_contig: int = 0, group=None) -> int:
    """choose_window_size over the WHOLE job (separate_reads.cpp:1466-1498): it looks at every read of every contig, so the
    ranks exchange (sum of lengths, reads, reads above 4 kb [, longest contig]) -- one tiny all-reduce, once per job. READ
    limits are (POS-1, POS + reference span) (input_output.cpp:503-511); the reference's `int sumLength` wraps."""
    import torch
    import torch.distributed as dist
    lens = np.asarray(local_rec_refspan, dtype=np.int64) + 2
    v = torch.tensor([int(lens.sum()), int(lens.size), int((lens > 4000).sum())], dtype=torch.int64)
    mx = torch.tensor([int(longest_contig)], dtype=torch.int64)
    if dist.is_available() and dist.is_initialized():
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        v = v.to(dev); mx = mx.to(dev)
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        v = v.cpu(); mx = mx.cpu()
    if amplicon:
        return int(mx.item())
    total, n, above = int(v[0]), int(v[1]), int(v[2])
    if n == 0:
        return 2000
    total &= 0xFFFFFFFF
    if total >= 1 << 31:
        total -= 1 << 32
    mean = total / float(n)
    if above < 20 and 2000 < mean < 4000:
        return 1000
    if above < 20 and mean < 2000:
        return 500
    return 2000


def mean_of_positive_f32(values) -> float:
    """call_variants.cpp:1312-1315,1377: float32 running sum of the positive entries in index order / their count.
    np.cumsum accumulates left to right in the array's dtype, i.e. exactly that running sum."""
    v = np.asarray(values, dtype=np.float32)
    v = v[v > 0]
    if v.size == 0:
        return float("nan")
    return float(np.float32(np.cumsum(v, dtype=np.float32)[-1] / np.float32(v.size)))


def gather_capacity(local_n: int, group=None) -> int:
    """One-off (outside the timed steps): the largest per-rank label count, so that the per-step gather is a single
    fixed-size collective with the actual count carried in-band."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return int(local_n)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(local_n)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def gather_labels(labels: np.ndarray, group=None, dst: int = 0, capacity: int = None):
    """The single gather of partition labels at the end (int16 on the wire: labels are -2, -1 or a group id < N).
    One collective: every rank sends `capacity` int16 labels preceded by its own count (two int16 words).
    Returns the list of per-rank label arrays on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [labels]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    assert labels.size == 0 or (labels.min() >= -2 and labels.max() < 32767)
    if capacity is None:
        capacity = gather_capacity(labels.size, group)
    assert labels.size <= capacity and labels.size < (1 << 30)
    # int16 labels shipped as raw bytes: neither NCCL/RCCL nor gloo has an int16 datatype
    buf16 = torch.full((capacity + 2,), -2, dtype=torch.int16)
    buf16[0] = labels.size & 0x7fff
    buf16[1] = labels.size >> 15
    if labels.size:
        buf16[2:2 + labels.size] = torch.from_numpy(labels.astype(np.int16))
    buf = buf16.view(torch.uint8).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return None
    res = []
    for o in out:
        v = o.cpu().view(torch.int16).numpy()
        n = int(v[0]) | (int(v[1]) << 15)
        res.append(v[2:2 + n].astype(np.int32))
    return res


class LabelGatherer:
    """gather_labels() with everything allocated once (pinned staging buffer, device buffer, receive buffers on `dst`): the
    per-step cost is one int32 -> int16 pass over the labels, one host-to-device copy and the one collective."""

    def __init__(self, capacity: int, group=None, dst: int = 0):
        import torch
        import torch.distributed as dist
        self.capacity = int(capacity)
        self.group = group
        self.dst = dst
        self.active = dist.is_available() and dist.is_initialized()
        if not self.active:
            return
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        self.host16 = torch.full((self.capacity + 2,), -2, dtype=torch.int16)
        if self.dev == "cuda":
            self.host16 = self.host16.pin_memory()
        self.np16 = self.host16.numpy()
        self.dev_buf = torch.empty((self.capacity + 2) * 2, dtype=torch.uint8, device=self.dev)
        self.out = [torch.empty_like(self.dev_buf) for _ in range(self.world)] if self.rank == dst else None
        self.copied = torch.cuda.Event() if self.dev == "cuda" else None   # the staging buffer is rewritten by the next step

    def gather(self, labels: np.ndarray, decode: bool = True):
        """Returns the per-rank label arrays on `dst` (raw device byte buffers if decode=False), None elsewhere."""
        import torch.distributed as dist
        if not self.active:
            return [labels]
        import torch
        n = int(labels.size)
        if n > self.capacity:
            raise ValueError(f"{n} labels do not fit the gather buffer of {self.capacity}")
        if n != getattr(self, "checked_n", None):    # int16 on the wire: checked on the first gather and whenever the job changes
            if n and not (int(labels.min()) >= -2 and int(labels.max()) < 32767):
                raise ValueError("a partition label does not fit the 16 bits it travels in")
            self.checked_n = n
        if self.copied is not None:
            self.copied.synchronize()      # the previous step's host-to-device copy has read the staging buffer
        self.np16[0] = n & 0x7fff
        self.np16[1] = n >> 15
        np.copyto(self.np16[2:2 + n], labels, casting="unsafe")      # labels are -2, -1 or a group id < 32767
        self.dev_buf.copy_(self.host16.view(dtype=torch.uint8), non_blocking=True)
        if self.copied is not None:
            self.copied.record()
        dist.gather(self.dev_buf, self.out, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        if not decode:
            return self.out
        res = []
        for o in self.out:
            v = o.cpu().view(__import__("torch").int16).numpy()
            k = int(v[0]) | (int(v[1]) << 15)
            res.append(v[2:2 + k].astype(np.int32))
        return res


# ---------------------------------------------------------------------------------------------
# The gather of a step in the form the .gro needs (separate_reads.cpp:1756-1784): per window the reads it holds and their
# labels. The final labels of a window come from three places (the device's cluster chains, windows without a seeding SNP:
# every read -1, windows without SNPs: the reads over the midpoint), so they are put together on the host; what crosses to
# the device for the collective is that list -- 6 bytes per (window, read) pair, an eighth of the dense [window][N reads of the
# contig] array of the round before -- staged in ONE pinned buffer that is allocated with the gatherer, copied once, gathered once.
# Payload (bytes): int64 n_windows, int64 n_rows (bit 62 set: with contigs) | int32 row_off[n_windows + 1] | [int32 win_contig[n_windows]] |
#                  int32 ids[n_rows] | int16 labels[n_rows]
# win_contig (optional): the contig of the JOB every window belongs to -- what whoever writes the .gro needs to put a rank's GROUP lines
# under the right CONTIG line, and what makes the digest below notice two windows that changed places.
# ---------------------------------------------------------------------------------------------
_WITH_CONTIGS = 1 << 62


def sparse_payload_bytes(n_windows: int, n_rows: int, with_contigs: bool = False) -> int:
    return 16 + 4 * (n_windows + 1) + (4 * n_windows if with_contigs else 0) + 4 * n_rows + 2 * n_rows + 8


def encode_sparse(win_row_off: np.ndarray, ids: np.ndarray, labels: np.ndarray, out: np.ndarray, win_contig: Optional[np.ndarray] = None) -> int:
    """Writes the payload into the uint8 array `out`; returns its length in bytes"""
    W, R = int(win_row_off.size) - 1, int(ids.size)
    n = sparse_payload_bytes(W, R, win_contig is not None)
    if out.size < n:
        raise ValueError(f"label payload of {n} bytes does not fit the gather buffer of {out.size} (the job's lists grew: size the SparseLabelGatherer again)")
    if R >= (1 << 31):
        raise ValueError("more than 2^31 (window, read) pairs in one rank's list")
    if R and (int(labels.min()) < -32768 or int(labels.max()) > 32767):
        raise ValueError("a partition label does not fit the 16 bits it travels in")
    if win_contig is not None and int(win_contig.size) != W:
        raise ValueError("one contig per window")
    out[:16].view(np.int64)[:] = (W, R | (_WITH_CONTIGS if win_contig is not None else 0))
    o = 16
    np.copyto(out[o:o + 4 * (W + 1)].view(np.int32), win_row_off, casting="unsafe"); o += 4 * (W + 1)
    if win_contig is not None:
        np.copyto(out[o:o + 4 * W].view(np.int32), win_contig, casting="unsafe"); o += 4 * W
    np.copyto(out[o:o + 4 * R].view(np.int32), ids, casting="unsafe"); o += 4 * R
    np.copyto(out[o:o + 2 * R].view(np.int16), labels, casting="unsafe")      # labels are -2, -1 or a group id < 32767
    return n


def decode_sparse(buf: np.ndarray):
    """(win_row_off int64, ids int32, labels int32[, win_contig int32]) of one rank's payload"""
    W, R = (int(x) for x in buf[:16].view(np.int64))
    with_contigs = bool(R & _WITH_CONTIGS)
    R &= ~_WITH_CONTIGS
    o = 16
    off = buf[o:o + 4 * (W + 1)].view(np.int32).astype(np.int64); o += 4 * (W + 1)
    wc = None
    if with_contigs:
        wc = buf[o:o + 4 * W].view(np.int32).copy(); o += 4 * W
    ids = buf[o:o + 4 * R].view(np.int32).copy(); o += 4 * R
    lab = buf[o:o + 2 * R].view(np.int16).astype(np.int32)
    return (off, ids, lab) if wc is None else (off, ids, lab, wc)


def sparse_digest(payloads):
    """Order-independent digest of per-window (reads, labels) lists, one (win_row_off, ids, labels[, win_contig]) tuple per rank: the number
    of windows and entries and the sum of the windows' CRC-32 -- over the window's contig (when the payload names it), its reads and their
    labels: the same job gives the same digest however its contigs were sharded over the ranks (a window's reads are indices within its
    contig), and two windows that changed places between contigs do not."""
    import zlib
    W = R = 0
    acc = 0
    for pay in payloads:
        off, ids, lab = pay[0], pay[1], pay[2]
        wc = np.ascontiguousarray(pay[3], dtype=np.int32) if len(pay) > 3 and pay[3] is not None else None
        off = np.asarray(off, dtype=np.int64); ids = np.ascontiguousarray(ids, dtype=np.int32); lab = np.ascontiguousarray(lab, dtype=np.int32)
        W += len(off) - 1; R += int(off[-1]) if len(off) else 0
        for w in range(len(off) - 1):
            a, b = int(off[w]), int(off[w + 1])
            seed = zlib.crc32(wc[w:w + 1].tobytes()) if wc is not None else 0
            acc = (acc + zlib.crc32(lab[a:b].tobytes(), zlib.crc32(ids[a:b].tobytes(), seed))) & 0xFFFFFFFFFFFFFFFF
    return {"windows": int(W), "entries": int(R), "sum_crc32": int(acc)}


class SparseLabelGatherer:
    """ONE gather per step of the ranks' (window, read, label) lists to `dst`. Everything is allocated here, once, for `capacity`
    bytes per rank (the largest payload of any rank: exchanged when the job is set up, not per step): the pinned staging buffer,
    the device buffer the collective sends, the receive buffers on `dst`."""

    def __init__(self, capacity_bytes: int, group=None, dst: int = 0):
        import torch
        import torch.distributed as dist
        self.capacity = (int(capacity_bytes) + 15) & ~15
        self.group, self.dst = group, dst
        self.active = dist.is_available() and dist.is_initialized()
        self.host = torch.zeros(self.capacity, dtype=torch.uint8)
        if not self.active:
            self.np = self.host.numpy()
            return
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        if self.dev == "cuda":
            self.host = self.host.pin_memory()
        self.np = self.host.numpy()
        self.dev_buf = torch.empty(self.capacity, dtype=torch.uint8, device=self.dev)
        self.out = [torch.empty_like(self.dev_buf) for _ in range(self.world)] if self.rank == dst else None
        self.copied = torch.cuda.Event() if self.dev == "cuda" else None   # the staging buffer is rewritten by the next step

    @staticmethod
    def job_capacity(local_bytes: int, group=None) -> int:
        """set-up: the largest payload over the ranks (+ 1/8: the lists of a job vary little from step to step, none at all for the same input)"""
        return gather_capacity(int(local_bytes) + int(local_bytes) // 8 + 64, group)

    def gather(self, win_row_off: np.ndarray, ids: np.ndarray, labels: np.ndarray, decode: bool = True, win_contig: Optional[np.ndarray] = None,
               to_host: bool = False):
        """Per-rank (win_row_off, ids, labels[, win_contig]) on `dst`, None elsewhere. decode=False: the raw payloads -- the collective's own
        receive buffers, or, with to_host, uint8 arrays over ONE pinned host block the payloads have been copied into when the call
        returns (what a single process ends its step with: the lists on the host)"""
        import torch.distributed as dist
        if self.active and self.copied is not None:
            self.copied.synchronize()      # the previous step's host-to-device copy has read the staging buffer
        n = encode_sparse(win_row_off, ids, labels, self.np, win_contig)
        if not self.active:
            return [decode_sparse(self.np[:n])] if decode else [self.host]
        self.dev_buf.copy_(self.host, non_blocking=True)
        if self.copied is not None:
            self.copied.record()
        dist.gather(self.dev_buf, self.out, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        if to_host or decode:
            import torch
            if self.dev == "cuda":
                if getattr(self, "host_out", None) is None:
                    self.host_out = torch.empty((self.world, self.capacity), dtype=torch.uint8).pin_memory()
                    self.landed = torch.cuda.Event()
                for k, o in enumerate(self.out):
                    self.host_out[k].copy_(o, non_blocking=True)
                self.landed.record()
                self.landed.synchronize()
                arrs = [self.host_out[k].numpy() for k in range(self.world)]
            else:
                arrs = [o.numpy() for o in self.out]
            return [decode_sparse(a) for a in arrs] if decode else arrs
        return self.out

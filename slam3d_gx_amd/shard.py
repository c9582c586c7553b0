"""Batch-mode sharding of independent frame pairs over ranks + the single gather of pose records.

The path partitions by frame pair: pair i of a batch of B goes to rank
i // (B / world) (contiguous blocks, so the all-gather order is the pair order).  No data-path
collective exists; the only exchange is one all-gather of 160-byte pose records per step
(RCCL over xGMI on the GPU box: backend "nccl"; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

RECORD_DOUBLES = 20   # 16 T (row-major) + norm + inliers + status + rmse  = 160 bytes


def shard_range(n_pairs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of rank `rank`; remainders go to the lowest ranks."""
    if world <= 0 or not (0 <= rank < world) or n_pairs < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_pairs, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pack_records(results: Sequence[dict]) -> np.ndarray:
    rec = np.zeros((len(results), RECORD_DOUBLES), dtype=np.float64)
    for i, r in enumerate(results):
        rec[i, :16] = np.asarray(r["T"], dtype=np.float64).reshape(16)
        rec[i, 16] = r["norm"]; rec[i, 17] = r["inliers"]; rec[i, 18] = r["status"]; rec[i, 19] = r.get("rmse", 0.0)
    return rec


def unpack_records(rec: np.ndarray) -> List[dict]:
    return [dict(T=r[:16].reshape(4, 4).copy(), norm=float(r[16]), inliers=int(r[17]), status=int(r[18]), rmse=float(r[19]))
            for r in np.asarray(rec).reshape(-1, RECORD_DOUBLES)]


def gather_records(local: np.ndarray, n_pairs: int, device=None):
    """All-gather the per-rank record blocks into the (n_pairs, 20) table, in pair order, on every rank.
    Ranks may hold different block sizes (remainder); blocks are padded to the largest."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(local).reshape(-1, RECORD_DOUBLES)
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_pairs, world, r) for r in range(world)]
    maxn = max(e - b for b, e in sizes)
    buf = torch.zeros((maxn, RECORD_DOUBLES), dtype=torch.float64, device=device)
    mine = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64).reshape(-1, RECORD_DOUBLES))
    buf[: mine.shape[0]].copy_(mine)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    parts = [o[: e - b].cpu().numpy() for o, (b, e) in zip(out, sizes)]
    return np.concatenate(parts, axis=0)


class PoseGatherer:
    """Per-step all-gather of the pose records with everything allocated once, pipelined by one step.

    `submit(local)` enqueues the all-gather of this step's records on a side stream (RCCL) and returns
    immediately, so the exchange overlaps the next step's kernels; `collect()` returns the table of the
    oldest submitted step (pair order, on every rank).  With gloo/CPU tensors the same calls run
    synchronously.  Exactly one collective per step; no data-path collective."""

    def __init__(self, n_pairs: int, device=None, force: bool = False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.n_pairs = n_pairs
        # force: run the collective even with one rank (developer knob: RCCL path on a single GPU)
        self.on = dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.world = dist.get_world_size() if self.on else 1
        self.rank = dist.get_rank() if self.on else 0
        self.sizes = [shard_range(n_pairs, self.world, r) for r in range(self.world)]
        self.maxn = max(1, max(e - b for b, e in self.sizes))
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.pending = []            # (work, slot)
        if not self.on:
            return
        kw = dict(dtype=torch.float64, device=device)
        self.nslots = 2              # a step's buffers are reused two submits later
        self.send = [torch.zeros((self.maxn, RECORD_DOUBLES), **kw) for _ in range(self.nslots)]
        self.recv = [torch.zeros((self.world * self.maxn, RECORD_DOUBLES), **kw) for _ in range(self.nslots)]
        pin = dict(dtype=torch.float64, pin_memory=self.cuda)
        self.h_send = [torch.zeros((self.maxn, RECORD_DOUBLES), **pin) for _ in range(self.nslots)]
        self.h_recv = [torch.zeros((self.world * self.maxn, RECORD_DOUBLES), **pin) for _ in range(self.nslots)]
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.done = [torch.cuda.Event() for _ in range(self.nslots)] if self.cuda else None
        self.k = 0

    def submit(self, local: np.ndarray) -> None:
        local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1, RECORD_DOUBLES)
        if not self.on:
            self.pending.append((None, local.copy()))
            return
        if len(self.pending) >= self.nslots:
            raise RuntimeError("collect() the previous step before submitting a third one")
        slot = self.k % self.nslots
        self.k += 1
        self.h_send[slot][: local.shape[0]].copy_(self.torch.from_numpy(local))
        if self.cuda:
            with self.torch.cuda.stream(self.stream):
                self.send[slot].copy_(self.h_send[slot], non_blocking=True)
                self.dist.all_gather_into_tensor(self.recv[slot], self.send[slot])
                self.h_recv[slot].copy_(self.recv[slot], non_blocking=True)
                self.done[slot].record(self.stream)
        else:
            self.send[slot].copy_(self.h_send[slot])
            self.dist.all_gather_into_tensor(self.recv[slot], self.send[slot])
            self.h_recv[slot].copy_(self.recv[slot])
        self.pending.append((slot, None))

    def collect(self) -> np.ndarray:
        slot, local = self.pending.pop(0)
        if not self.on:
            return local
        if self.cuda:
            self.done[slot].synchronize()
        t = self.h_recv[slot].numpy().reshape(self.world, self.maxn, RECORD_DOUBLES)
        return np.concatenate([t[r, : e - b] for r, (b, e) in enumerate(self.sizes)], axis=0)

    def gather(self, local: np.ndarray) -> np.ndarray:
        """Unpipelined form: submit + collect."""
        self.submit(local)
        return self.collect()


def dense_row_range(height: int, world: int, rank: int) -> Tuple[int, int]:
    """Dense mode (one pair over all ranks): source image rows of this rank."""
    return shard_range(height, world, rank)

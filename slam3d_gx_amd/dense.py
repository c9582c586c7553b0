"""Dense mode (BASELINE config 5): ONE frame pair sharded over ranks by source rows.

Every rank holds the full target (tiles, boxes, normals) and the source rows
shard.dense_row_range(height, world, rank).  Per iteration each rank runs the NN search and the
normal-equation accumulation on its rows (slam3d_icp_dense_partial), the 29 partial sums are
all-reduced (the path's single exchange step: 232 bytes), and every rank solves the same 6x6 /
3x3 system and updates T identically (slam3d_icp_dense_update).  Correspondences are unaffected by
the sharding (each query still sees the whole target), and because the sums are int64 fixed point
(integer addition is associative) the pose is bit-identical to the 1-rank run for any sharding.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from . import shard


def allreduce_sum_torch(device=None) -> Callable[[np.ndarray], np.ndarray]:
    """all-reduce (SUM) of the 29 int64 fixed-point sums over torch.distributed (RCCL on the GPU box, gloo in CPU tests)."""
    import torch
    import torch.distributed as dist

    def f(x: np.ndarray) -> np.ndarray:
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return x
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.int64))
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    return f


def dense_align(handle, world: int, rank: int, T_init=None, allreduce: Optional[Callable] = None, stream: int = 0) -> dict:
    """Runs params.iterations dense iterations on `handle` (slot 0 must hold the pair)."""
    r0, r1 = shard.dense_row_range(handle.params.height, world, rank)
    handle.dense_set_rows(r0, r1)
    handle.dense_begin(T_init, stream)
    total = np.zeros(29, dtype=np.int64)
    for _ in range(handle.params.iterations):
        part = handle.dense_partial(stream)
        total = allreduce(part) if allreduce is not None else part
        handle.dense_update(total, stream)
    res = handle.dense_finish(total)
    handle.dense_set_rows(0, handle.params.height)
    return res


def dense_align_device(handle, world: int, rank: int, d_sums, T_init=None, stream: int = 0) -> dict:
    """dense_align with the exchange kept on the device: `d_sums` is a torch int64 tensor of 29 elements on the
    handle's GPU; per iteration partial -> in-place all-reduce (RCCL, same stream) -> update, with no host
    synchronisation until the final fetch."""
    import torch.distributed as dist
    r0, r1 = shard.dense_row_range(handle.params.height, world, rank)
    handle.dense_set_rows(r0, r1)
    handle.dense_begin(T_init, stream)
    on = dist.is_initialized() and dist.get_world_size() > 1
    ptr = d_sums.data_ptr()
    for _ in range(handle.params.iterations):
        handle.dense_partial_device(ptr, stream)
        if on:
            dist.all_reduce(d_sums, op=dist.ReduceOp.SUM)
        handle.dense_update_device(ptr, stream)
    res = handle.dense_finish_device(ptr, stream)
    handle.dense_set_rows(0, handle.params.height)
    return res

"""Dense mode (BASELINE config 5): ONE frame pair sharded over ranks by source rows.

Every rank holds the full target (tiles, boxes, normals) and the source rows
shard.dense_row_range(height, world, rank).  Per iteration each rank runs the NN search and the
normal-equation accumulation on its rows (slam3d_icp_dense_partial), the 36 partial Gram totals are
all-reduced (the path's single exchange step: 36 int64 = 288 bytes; slam3d_icp_dense_run exchanges the 16 x 40 int64 accumulator set in place), and every rank solves the same 6x6 /
3x3 system and updates T identically (slam3d_icp_dense_update).  Correspondences are unaffected by
the sharding (each query still sees the whole target), and because the sums are exact int64 Gram totals
(integer addition is associative) the pose is bit-identical to the 1-rank run for any sharding.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from . import shard


def allreduce_sum_torch(device=None) -> Callable[[np.ndarray], np.ndarray]:
    """all-reduce (SUM) of the 36 int64 Gram totals over torch.distributed (RCCL on the GPU box, gloo in CPU tests)."""
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
    total = np.zeros(36, dtype=np.int64)
    for _ in range(handle.params.iterations):
        part = handle.dense_partial(stream)
        total = allreduce(part) if allreduce is not None else part
        handle.dense_update(total, stream)
    res = handle.dense_finish(total)
    handle.dense_set_rows(0, handle.params.height)
    return res


def dense_align_device(handle, world: int, rank: int, d_sums, T_init=None, stream: int = None, force_collective: bool = False) -> dict:
    """dense_align with the exchange kept on the device, through torch.distributed: `d_sums` is a torch int64 tensor
    of 36 elements on the handle's GPU; per iteration partial -> in-place all-reduce -> update, with no host
    synchronisation until the final fetch.

    Stream contract (round-1 bug: the kernels went to the handle's private non-blocking stream while ProcessGroupNCCL
    ordered the all-reduce against torch's CURRENT stream, so the collective could read the sums before they were
    written).  The three steps must share one stream that is also torch's current stream: by default an explicit
    non-default torch stream is created, made current for the loop and passed down as the launch stream.  A caller
    stream is accepted when it is a real (non-null) stream; the null stream with more than one rank is refused.
    `slam3d_icp_dense_run` (capi.IcpHandle.dense_run) does the same loop inside the library with RCCL called from C
    and is what bench.py uses."""
    import torch
    import torch.distributed as dist
    on = dist.is_initialized() and (dist.get_world_size() > 1 or force_collective)
    if stream is not None and stream == 0 and (world > 1 or force_collective):
        raise ValueError("dense_align_device: the null stream cannot order the all-reduce with the handle's kernels; "
                         "pass a non-default stream (or None to let this function create one)")
    r0, r1 = shard.dense_row_range(handle.params.height, world, rank)
    handle.dense_set_rows(r0, r1)
    side = None
    if stream is None:
        side = torch.cuda.Stream(device=d_sums.device)
        side.wait_stream(torch.cuda.current_stream(d_sums.device))
        stream = side.cuda_stream
    if side is not None:
        ctx = torch.cuda.stream(side)
    elif stream == 0:
        import contextlib
        ctx = contextlib.nullcontext()          # one rank, no collective: the legacy default stream orders nothing else
    else:
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=d_sums.device))
    ptr = d_sums.data_ptr()
    with ctx:
        handle.dense_begin(T_init, stream)
        for _ in range(handle.params.iterations):
            handle.dense_partial_device(ptr, stream)
            if on:
                dist.all_reduce(d_sums, op=dist.ReduceOp.SUM)       # enqueued on the current stream == `stream`
            handle.dense_update_device(ptr, stream)
        res = handle.dense_finish_device(ptr, stream)
    handle.dense_set_rows(0, handle.params.height)
    return res


def host_staged_allreduce():
    """An all-reduce for IcpHandle.dense_run_with that needs no RCCL: drain the stream, stage the int64 buffer through host memory,
    SUM it with torch.distributed (gloo works with several ranks on ONE GPU -- RCCL needs a GPU per rank), write it back.  The
    exchange is then synchronous; it exists to run the library's dense loop and its failure protocol with two ranks in tests."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    hip = C.CDLL("libamdhip64.so")

    def f(d_buf: int, count: int, stream: int) -> int:
        if hip.hipStreamSynchronize(C.c_void_p(stream)) != 0:
            return 1
        host = np.empty(count, dtype=np.int64)
        if hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(d_buf), C.c_size_t(count * 8), C.c_int(2)) != 0:      # hipMemcpyDeviceToHost
            return 1
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if hip.hipMemcpy(C.c_void_p(d_buf), C.c_void_p(host.ctypes.data), C.c_size_t(count * 8), C.c_int(1)) != 0:      # hipMemcpyHostToDevice
            return 1
        return 0

    return f

"""N>1 path on CPU: world_size-2 gloo run of the pair sharding + pose-record all-gather
(the same slam3d_gx_amd.shard code bench.py runs over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from slam3d_gx_amd import shard, synth


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 512, 513):
        for w in (1, 2, 3, 4, 8):
            rs = [shard.shard_range(n, w, r) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [e - b for b, e in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(4, 2, 2)


def test_record_roundtrip():
    res = [dict(T=np.arange(16.0).reshape(4, 4) + i, norm=0.1 * i, inliers=100 + i, status=i % 3, rmse=0.01 * i) for i in range(5)]
    back = shard.unpack_records(shard.pack_records(res))
    assert shard.pack_records(res).nbytes == 5 * 160
    for a, b in zip(res, back):
        assert np.array_equal(a["T"], b["T"]) and a["inliers"] == b["inliers"] and a["status"] == b["status"]


SEEDS = [1000, 1001, 1002, 1003, 1004]   # 5 pairs over 2 ranks: uneven blocks


def _solve_pair(seed):
    pr = synth.make_pair(seed, 64, 48)
    s4 = synth.backproject_numpy(pr.depth_src, pr.intr)
    t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
    return O.icp(s4, t4, O.params(pr.intr, iterations=2, estimator=1, threads=1), trace=False)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard.shard_range(len(SEEDS), world, rank)
    local = [_solve_pair(s) for s in SEEDS[b:e]]
    table = shard.gather_records(shard.pack_records(local), len(SEEDS))
    if rank == 0:
        q.put(table)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    table = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = shard.pack_records([_solve_pair(s) for s in SEEDS])
    assert table.shape == (len(SEEDS), shard.RECORD_DOUBLES)
    assert np.array_equal(table, ref)

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
    # the preallocated, one-step-pipelined gatherer bench.py uses must give the same table
    pg = shard.PoseGatherer(len(SEEDS))
    rec = shard.pack_records(local)
    pg.submit(rec)
    pg.submit(rec * 2.0)
    t1, t2 = pg.collect(), pg.collect()
    assert np.array_equal(t1, table) and np.array_equal(t2, table * 2.0)
    assert np.array_equal(pg.gather(rec), table)
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


def _dense_worker(rank, world, port, q):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from slam3d_gx_amd import dense

    class FakeHandle:
        """stands in for capi.IcpHandle: partial sums are a deterministic function of (rows, iteration)"""
        class P:
            height = 48; iterations = 3
        params = P()

        def __init__(self):
            self.rows = None; self.it = 0; self.updates = []
        def dense_set_rows(self, a, b): self.rows = (a, b)
        def dense_begin(self, T, stream): self.it = 0
        def dense_partial(self, stream):
            a, b = self.rows
            return np.arange(29, dtype=np.int64) * (b - a) + 1000 * self.it + a
        def dense_update(self, total, stream): self.updates.append(total.copy()); self.it += 1
        def dense_finish(self, total): return dict(total=total.copy(), updates=self.updates)

    h = FakeHandle()
    out = dense.dense_align(h, world, rank, allreduce=dense.allreduce_sum_torch())
    q.put((rank, out["total"], h.rows))
    dist.barrier()
    dist.destroy_process_group()


def test_dense_allreduce_loop_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dense_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # rows 0..24 and 24..48; last iteration it=2: sum over ranks of (k*24 + 2000 + a)
    want = np.arange(29, dtype=np.int64) * 48 + 4000 + 24
    assert np.array_equal(got[0][1], want) and np.array_equal(got[1][1], want)
    assert got[0][2] == (0, 48) and got[1][2] == (0, 48)      # rows restored after the run

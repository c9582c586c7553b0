"""Worker of tests/test_two_gpus.py: one process per GPU (launched through torch.distributed.run).  Every rank runs BASELINE
config 5's dense loop over a real multi-rank RCCL communicator and compares it bit for bit with its own unsharded run; the
pose gather must come back in rank order."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from slam3d_gx_amd import capi, synth
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("gloo", rank=rank, world_size=world)          # host side only: the 128-byte id travels through it
    uid = [capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    comm = capi.Comm(uid[0], rank, world, local)
    pr = synth.make_pair(2001, 320, 240)
    s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
    params = capi.default_params(pr.intr, iterations=10, device=local)
    ok = True
    for head in ("1", "0"):                       # one exchange per iteration (head solve) and the three-step form
        os.environ["SLAM3D_HEAD_SOLVE"] = head
        with capi.IcpHandle(params) as h:
            h.set_clouds_host(0, s4, t4)
            alone = h.dense_run(None)
            T_alone = h.get_trace(0)[0].copy()
            h.set_clouds_host(0, s4, t4)
            shared = h.dense_run(comm)
            T_shared = h.get_trace(0)[0].copy()
        ok = ok and np.array_equal(alone["T_raw"], shared["T_raw"]) and np.array_equal(T_alone, T_shared)
        ok = ok and alone["inliers"] == shared["inliers"] and alone["status"] == shared["status"] and alone["rmse"] == shared["rmse"]
        ok = ok and shared["n_src"] < alone["n_src"]                      # this rank really held only its rows
    recs = [dict(T=np.eye(4) * (10 * rank + k + 1), norm=float(rank), inliers=100 * rank + k, status=0, rmse=0.0) for k in range(3)]
    table = comm.gather(recs)
    ok = ok and len(table) == 3 * world and all(int(table[3 * r + k]["inliers"]) == 100 * r + k for r in range(world) for k in range(3))
    comm.close()
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    dist.destroy_process_group()
    if rank == 0:
        print("TWO_GPU_RESULT", all(flags), flush=True)
    sys.exit(0 if all(flags) else 1)


if __name__ == "__main__":
    main()

"""Worker of tests/test_frames_comm.py::test_dense_failure_injection_two_ranks: one rank of a two-rank dense alignment on ONE GPU,
exchange through gloo (slam3d_icp_dense_run_with + dense.host_staged_allreduce).  argv: rank world port estimator fail_at out.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

rank, world, port, est, fail_at, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = port
import torch.distributed as dist
from slam3d_gx_amd import capi, dense, synth

dist.init_process_group("gloo", rank=rank, world_size=world)
pr = synth.make_pair(1000, 320, 240)
s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
res = {"rank": rank}
with capi.IcpHandle(capi.default_params(pr.intr, iterations=8, estimator=est)) as h:
    h.set_clouds_host(0, s4, t4)
    if fail_at != -1 and rank == 1:
        h.set_fault_injection(fail_at)                          # only THIS rank's handle fails
    try:
        r = h.dense_run_with(rank, world, dense.host_staged_allreduce())
        res.update(code=0, status=r["status"], T=np.asarray(r["T_raw"]).reshape(16).tolist(), inliers=r["inliers"])
    except capi.Slam3dError as e:
        res.update(code=e.code, message=str(e))
json.dump(res, open(out, "w"))
dist.barrier()
dist.destroy_process_group()

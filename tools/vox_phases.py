#!/usr/bin/env python3
"""Developer tool (MI355X): phases of k_voxel_insert per block, from a -DVOX_DBG build (SLAM3D_LIB=tools/variants/voxdbg.so)."""
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from slam3d_gx_amd import capi, synth
pr = synth.make_pair(1000, 640, 480)
c = synth.backproject_numpy(pr.depth_src, pr.intr).reshape(-1, 4).copy()
c[:, 3] = np.random.default_rng(0).integers(0, 2 ** 32, c.shape[0], dtype=np.uint64).astype(np.uint32).view(np.float32)
dev = torch.device("cuda:0")
d = torch.from_numpy(c).to(dev); out = torch.zeros_like(d)
h = capi.IcpHandle(capi.default_params(pr.intr, max_batch=1, device=0))
s = torch.cuda.Stream(device=dev)
for _ in range(5):
    m = h.voxel_grid_device(d.data_ptr(), len(c), out.data_ptr(), 0.03, s.cuda_stream)
torch.cuda.synchronize()
lib = capi.load_library()
buf = (ctypes.c_longlong * (1200 * 8))()
rc = lib.slam3d_debug_vox_phases(buf, 1200 * 8)
a = np.array(buf, dtype=np.int64).reshape(1200, 8)
t0 = a[:, 0].min()
print("voxels", m, "rc", rc)
names = ["lds init+sync->load", "scans", "lds atomics+sync", "compact+sync", "CAS", "stores/adds+sync", "drain vmcnt"]
dd = np.diff(a, axis=1) / 100.0
for k, nme in enumerate(names):
    print(f"{nme:24s} median {np.median(dd[:, k]):6.2f}  p90 {np.percentile(dd[:, k], 90):6.2f}  max {dd[:, k].max():6.2f} us")
print("block start (us after first): median %.2f max %.2f; block end: median %.2f max %.2f" % (np.median(a[:, 0] - t0) / 100, (a[:, 0] - t0).max() / 100,
      np.median(a[:, 7] - t0) / 100, (a[:, 7] - t0).max() / 100))
print("block lifetime median %.2f max %.2f" % (np.median(a[:, 7] - a[:, 0]) / 100, (a[:, 7] - a[:, 0]).max() / 100))

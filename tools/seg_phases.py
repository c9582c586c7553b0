"""Developer tool (MI355X): phases of k_seg_count per block and RANSAC round, from a -DSEGC_DBG build (SLAM3D_LIB=tools/variants/segdbg.so)."""
import os, sys, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam3d_gx_amd import capi, synth
pr = synth.make_pair(1000, 640, 480)
c = synth.backproject_numpy(pr.depth_src, pr.intr)
with capi.IcpHandle(capi.default_params(pr.intr, max_batch=1)) as h:
    for _ in range(3):
        planes, labels = h.segment_planes(c)
lib = capi.load_library()
buf = (ctypes.c_longlong * (3 * 1200 * 8))()
lib.slam3d_debug_segc_phases(buf, 3 * 1200 * 8)
a = np.array(buf, dtype=np.int64).reshape(3, 1200, 8)
names = ["pixel loads issued + head (draws)", "sync", "consensus loop", "LDS atomics + sync", "global atomics drained"]
for r in range(3):
    x = a[r][:, :6]
    t0 = x[:, 0].min()
    print(f"round {r}: block start after first: median {np.median(x[:,0]-t0)/100:.2f} max {(x[:,0]-t0).max()/100:.2f} us; block end: median {np.median(x[:,5]-t0)/100:.2f} max {(x[:,5]-t0).max()/100:.2f} us")
    d = np.diff(x, axis=1) / 100.0
    for k, n in enumerate(names):
        print(f"   {n:36s} median {np.median(d[:,k]):5.2f} p90 {np.percentile(d[:,k],90):5.2f} max {d[:,k].max():5.2f}")

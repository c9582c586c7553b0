#!/usr/bin/env python3
"""PCIe-inclusive rate of the one-call API (host buffers handed over at the boundary) -- DESIGN.md section 7."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam3d_gx_amd import capi, synth

pr = synth.make_pair(1000)
s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
s8 = np.zeros(s4.shape[:2] + (8,), np.float32); s8[..., :3] = s4[..., :3]
t8 = np.zeros(t4.shape[:2] + (8,), np.float32); t8[..., :3] = t4[..., :3]
with capi.IcpHandle(capi.default_params(pr.intr, iterations=20)) as h:
    for name, fn in (("host u16 depth (2 x 0.6 MB)", lambda: h.align_depth_batch([pr.depth_src], [pr.depth_tgt])),
                     ("host float4 clouds (2 x 4.9 MB)", lambda: h.align(s4, t4)),
                     ("host PointXYZRGBA-stride clouds (2 x 9.8 MB)", lambda: h.align(s8, t8))):
        fn(); fn()
        t0 = time.perf_counter(); n = 10
        for _ in range(n): fn()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:45s} {1e3*dt:7.3f} ms/pair  {20/dt:9.1f} ICP iterations/s")

#!/usr/bin/env python3
"""Developer tool: per-wave statistics of the tile-pruned NN kernel (needs SLAM3D_NN_DEBUG=1)."""
import os, sys
os.environ["SLAM3D_NN_DEBUG"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam3d_gx_amd import capi, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pr = synth.make_pair(1000)
s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
for it in (1, iters):
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=it)) as h:
        h.align(s4, t4)
        d = h.get_nn_debug()
    act = d[:, 4] > 0
    d = d[act]
    life = d[:, 4] - d[:, 0]
    print(f"--- last of {it} iteration(s): {act.sum()} active waves; kernel span {d[:,4].max()-d[:,0].min()} clk")
    d = d[d[:, 1] > 0]
    lo32 = lambda x: x & 0xffffffff
    hi32 = lambda x: x >> 32
    for name, v in (("lifetime", d[:, 4] - d[:, 0]), ("prologue", d[:, 1] - d[:, 0]), ("own 5 tiles", d[:, 2] - d[:, 1]), ("publish+items (to barrier 2)", d[:, 3] - d[:, 2]),
                    ("epilogue", d[:, 4] - d[:, 3]), ("tiles scanned", lo32(d[:, 5])), ("candidates", lo32(d[:, 6])), ("batches", lo32(d[:, 7])),
                    ("cells swept", hi32(d[:, 5])), ("fine hits", hi32(d[:, 6])), ("refined hits", hi32(d[:, 7]))):
        print(f"{name:14s} mean {v.mean():10.1f}  p50 {np.percentile(v,50):9.0f}  p90 {np.percentile(v,90):9.0f}  p99 {np.percentile(v,99):9.0f}  max {v.max():9.0f}")

#!/usr/bin/env python3
"""Developer tool: the unorganized-cloud ICP (16 k x 15 k voxel clouds of the reference's Kinect frames), kernel time by bucket.
usage: tools/quick_unorg.py ["ENV=1,..."] ..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from slam3d_gx_amd import capi, synth
import test_unorganized as U
v1, v2 = U.kinect_voxel_clouds()
W = 16384
intr = synth.Intrinsics(width=W, height=1)
a = np.ascontiguousarray(U.pad(v1, len(v1))); b = np.ascontiguousarray(U.pad(v2, len(v2)))
for cfg in (sys.argv[1:] or [""]):
    keys = []
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("="); os.environ[k] = v; keys.append(k)
    for mode in ((capi.NN_AUTO,) if os.environ.get("QU_AUTO_ONLY") else (capi.NN_AUTO, capi.NN_TILES, capi.NN_BRUTE_VALU)):
        with capi.IcpHandle(capi.default_params(intr, iterations=20, estimator=capi.EST_SVD, nn_mode=mode)) as h:
            for _ in range(3):
                h.align(a, b)
            t0 = time.perf_counter()
            for _ in range(20):
                h.set_clouds_host(0, a, b); h.run(1); h.fetch_results(1)
            wall = (time.perf_counter() - t0) / 20
            h.set_profiling(True)
            h.align(a, b)
            tm = h.get_timings(); its = 1e3 * h.get_iteration_timings()
        print(f"{cfg or 'default':20s} mode {mode} wall {1e3*wall:.3f} ms  pre {1e3*tm['preprocess_ms']:.0f} us nn {1e3*tm['nn_ms']:.0f} us rest {1e3*tm['accumulate_solve_ms']:.0f} us | nn per it " + " ".join(f"{x:.0f}" for x in its[:8]), flush=True)
    for k in keys:
        del os.environ[k]

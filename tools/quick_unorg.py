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
        if os.environ.get("SLAM3D_LIST_DEBUG") and mode == capi.NN_AUTO:
            with capi.IcpHandle(capi.default_params(intr, iterations=20, estimator=capi.EST_SVD, nn_mode=mode)) as h2:
                h2.align(a, b); h2.align(a, b)
                d = h2.get_list_debug()
            names = ["bounds", "listing", "scans", "rows", "gram", "arrive", "wait", "totals", "derive", "solve"]
            print("  per iteration, 10 ns ticks -> us; 'search' = bounds..gram of thread 0; slowest = the block with the longest search")
            print("  it | blocks | search median / p90 / max | slowest block: " + " ".join(names) + " | tiles scanned (wave 0) | iteration us")
            for it in range(d.shape[0]):
                live = d[it, :, 11] > 0
                ph = d[it, live, :10] * 0.01
                srch = ph[:, :5].sum(1)
                k = int(srch.argmax())
                nxt = d[it + 1, live, 11].min() if it + 1 < d.shape[0] else 0
                tot_us = (nxt - d[it, live, 11].min()) * 0.01 if nxt else float("nan")
                print(f"  {it:2d} | {live.sum():3d} | {np.median(srch):5.1f} {np.percentile(srch, 90):5.1f} {srch.max():5.1f} | " + " ".join(f"{x:5.1f}" for x in ph[k]) +
                      f" | {d[it, live, 10][k]:2d} | {tot_us:6.1f}")
            print("  mean over blocks and iterations: " + " ".join(f"{n} {x:.1f}" for n, x in zip(names, (d[:, :, :10][d[:, :, 11] > 0] * 0.01).mean(0))))
        print(f"{cfg or 'default':20s} mode {mode} wall {1e3*wall:.3f} ms  pre {1e3*tm['preprocess_ms']:.0f} us nn {1e3*tm['nn_ms']:.0f} us rest {1e3*tm['accumulate_solve_ms']:.0f} us | nn per it " + " ".join(f"{x:.0f}" for x in its[:8]), flush=True)
    for k in keys:
        del os.environ[k]

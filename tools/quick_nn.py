#!/usr/bin/env python3
"""Developer tool: NN launch time and un-overlapped run time of single 640x480 pairs under environment knobs.
usage: tools/quick_nn.py "SLAM3D_NN_GX=1280" "SLAM3D_NN_GX=1536,SLAM3D_XCD_BANDS=0" ...   (one line per configuration)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam3d_gx_amd import capi, synth

seeds = [1000, 1001, 1002, 1003, 1004, 1005]
prs = synth.make_pairs([(s, 640, 480, 0.0002) for s in seeds])
data = [(p.depth_src, p.depth_tgt) for p in prs.values()]
intr = list(prs.values())[0].intr
for cfg in (sys.argv[1:] or [""]):
    keys = []
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
        keys.append(k)
    with capi.IcpHandle(capi.default_params(intr, iterations=20)) as h:
        for s, t in data[:2]:
            h.align_depth_batch([s], [t])
        t0 = time.perf_counter()
        n = 0
        for rep in range(5):
            for s, t in data:
                h.align_depth_batch([s], [t]); n += 1
        wall = (time.perf_counter() - t0) / n
        h.set_profiling(True)
        its, tot = [], []
        for rep in range(2):
            for s, t in data:
                h.align_depth_batch([s], [t])
                its.append(h.get_iteration_timings()); tot.append(h.get_timings()["total_ms"])
        its = np.array(its)
    print(f"{cfg or 'default':50s} wall {1e3*wall:7.3f} ms/pair  nn launch it3..19 {1e3*its[:,3:].mean():6.2f} us (min {1e3*its[:,3:].min():5.1f} max {1e3*its[:,3:].max():5.1f})  it0 {1e3*its[:,0].mean():5.1f} it1 {1e3*its[:,1].mean():5.1f} it2 {1e3*its[:,2].mean():5.1f}  profiled total {np.mean(tot):6.3f} ms", flush=True)
    for k in keys:
        del os.environ[k]

#!/usr/bin/env python3
"""Developer tool: NN launch time per iteration, one alignment at a time, on a synthetic pair and on the reference's Kinect
frames (dep_k -> dep_k from 2 deg / 3 cm; dep1 -> dep2), under environment knobs.
usage: tools/quick_real.py "" "SLAM3D_WIN_MIX=16" ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam3d_gx_amd import capi, synth
from PIL import Image
kin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kinect")
d1 = np.array(Image.open(os.path.join(kin, "exp1_dep_1.png"))).astype(np.uint16)
d2 = np.array(Image.open(os.path.join(kin, "exp1_dep_2.png"))).astype(np.uint16)
pr = synth.make_pair(1000)
Ti = synth.pose_from_seed(77, 2.0, 0.03)
prb = synth.make_pair(1000, noise_sigma=0.0012, hole_block=8, hole_prob=0.25)
cases = [("synthetic", pr.depth_src, pr.depth_tgt, None), ("baseline_md", prb.depth_src, prb.depth_tgt, None), ("dep1->dep1", d1, d1, Ti), ("dep2->dep2", d2, d2, Ti), ("dep1->dep2", d1, d2, None)]
for cfg in (sys.argv[1:] or [""]):
    keys = []
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("="); os.environ[k] = v; keys.append(k)
    for name, s, t, T0 in cases:
        with capi.IcpHandle(capi.default_params(synth.Intrinsics(), iterations=20, max_corr_dist=float(os.environ.get("QR_GATE", "0.10")),
                                                estimator=int(os.environ.get("QR_EST", "0")), plane_flags=int(os.environ.get("QR_FLAGS", "0")))) as h:
            kw = {} if T0 is None else {"T_init": T0.reshape(1, 16)}
            for _ in range(2):
                h.align_depth_batch([s], [t], **kw)
            h.set_profiling(True)
            its = []
            for _ in range(4):
                r = h.align_depth_batch([s], [t], **kw)
                its.append(h.get_iteration_timings())
            its = 1e3 * np.array(its).mean(axis=0)
            pre = h.get_timings()['preprocess_ms'] * 1e3
        print(f"{cfg or 'default':24s} {name:11s} pre {pre:5.0f} nn sum {its.sum():7.1f} us  per it: " + " ".join(f"{x:4.0f}" for x in its), flush=True)
    for k in keys:
        del os.environ[k]

#!/usr/bin/env python3
"""Developer tool: launch time and TFLOP/s of the two full-scan NN kernels (k_nn_mfma, k_nn_valu) under environment knobs.
usage: tools/quick_brute.py "" "SLAM3D_MFMA_SPLIT=16" ...   (one line per configuration)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam3d_gx_amd import capi, synth

pr = synth.make_pair(1000, 640, 480, **(dict(noise_sigma=0.0012, hole_block=8, hole_prob=0.25) if os.environ.get("QB_BMD") else {}))
s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
for cfg in (sys.argv[1:] or [""]):
    keys = []
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("="); os.environ[k] = v; keys.append(k)
    out = []
    for mode, name in ((capi.NN_BRUTE_MFMA, "mfma"), (capi.NN_BRUTE_VALU, "valu")):
        with capi.IcpHandle(capi.default_params(pr.intr, iterations=4, nn_mode=mode, coarse_iterations=0)) as h:
            h.set_clouds_host(0, s4, t4); h.set_profiling(True); h.run(1); h.fetch_results(1)
            ms = []
            for _ in range(3):
                h.set_clouds_host(0, s4, t4); h.run(1); r = h.fetch_results(1)[0]
                ms.append(float(np.mean(h.get_iteration_timings()[1:])))
            fl = 8.0 * r["n_src"] * r["n_tgt"]
            out.append(f"{name} {min(ms):6.3f} ms {fl / (min(ms) * 1e-3) / 1e12:6.1f} TF")
    print(f"{cfg or 'default':40s} " + "   ".join(out), flush=True)
    for k in keys:
        del os.environ[k]

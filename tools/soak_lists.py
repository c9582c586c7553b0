#!/usr/bin/env python3
"""Randomised parity soak of the POINT-LIST path (developer tool, needs an MI355X): random list sizes (1 .. 40 k points: below a wave, beyond
the 384 tiles whose boxes fit LDS), random subsets / shuffles of real and synthetic clouds, gates, iteration counts, coarse iterations, initial
poses, svd and planes-only estimators (with and without the optional gates): the persistent launch (k_list_icp) vs the oracle -- indices, d2,
every iterate T, the sums, status and inliers must be identical.
usage: tools/soak_lists.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import test_unorganized as U
from slam3d_gx_amd import capi, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
v1, v2 = U.kinect_voxel_clouds()
pools = [v1, v2]
for seed in (1000, 1001):            # dense synthetic frames as lists (up to ~60 k points)
    pr = synth.make_pair(seed, 320, 240)
    for d in (pr.depth_src, pr.depth_tgt):
        c = synth.backproject_numpy(d, pr.intr).reshape(-1, 4)
        pools.append(c[np.isfinite(c[:, 2])].copy())
bad = 0
t0 = time.time()
for case in range(n_cases):
    a = pools[int(rng.integers(0, len(pools)))]; b = pools[int(rng.integers(0, len(pools)))]
    if rng.random() < 0.5:
        b = a                                                   # a perturbed self-alignment
    na = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 200, 1000, 4000, 12000, len(a), 40000])); nb = int(rng.choice([1, 4, 64, 300, 2500, 9000, len(b), 40000]))
    na, nb = min(na, len(a)), min(nb, len(b))
    sa = a[rng.choice(len(a), na, replace=False)]; sb = b[rng.choice(len(b), nb, replace=False)]
    W = max(na, nb) + int(rng.integers(0, 100))
    intr = synth.Intrinsics(width=W, height=1)
    iters = int(rng.integers(1, 7)) if rng.random() < 0.4 else int(rng.integers(8, 26))
    gate = float(rng.choice([0.02, 0.05, 0.1, 0.3, 1.0]))
    coarse = int(rng.choice([0, 1, 3, 5]))
    Ti = synth.pose_from_seed(int(rng.integers(0, 1 << 20)), 3.0, 0.05) if rng.random() < 0.7 else None
    kw = dict(iterations=iters, max_corr_dist=gate, coarse_iterations=coarse)
    plane = rng.random() < 0.3 and min(na, nb) >= 300
    if plane:
        kw.update(estimator=capi.EST_PLANE, plane_flags=capi.PLANE_ONLY); okw = dict(kw, estimator=2, plane_only=1); okw.pop("plane_flags")
        if rng.random() < 0.5:
            g8 = dict(max_plane_residual2=float(rng.choice([1e-4, 4e-4])), min_normal_cos=float(rng.choice([0.0, 0.9])))
            kw.update(g8); okw.update(g8)
    else:
        kw.update(estimator=capi.EST_SVD); okw = dict(kw, estimator=1)
    src = np.ascontiguousarray(U.pad(sa, na)); tgt = np.ascontiguousarray(U.pad(sb, nb))
    ro = O.icp(U.pad(sa, W), U.pad(sb, W), O.params(intr, nn_method=0 if na * nb < 4e7 else 1, **okw), T_init=Ti)
    with capi.IcpHandle(capi.default_params(intr, **kw)) as h:
        rg = h.align(src, tgt, Ti)
        idx, d2 = h.get_correspondences(0)
        Tt, St = h.get_trace(0)
    ok = (np.array_equal(idx, ro["idx"]) and np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32))
          and np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:iters], ro["sums_trace"])
          and rg["status"] == ro["status"] and rg["inliers"] == ro["inliers"])
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, na=na, nb=nb, W=W, iters=iters, gate=gate, coarse=coarse, plane=plane, kw=kw), "idx", int((idx != ro["idx"]).sum()), flush=True)
print(f"{n_cases} list cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)

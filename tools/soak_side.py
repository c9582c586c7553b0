#!/usr/bin/env python3
"""Randomised parity soak for the f-rows (developer tool, needs an MI355X): plane segmentation and voxel grid vs their oracles."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from slam3d_gx_amd import capi, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for case in range(n_cases):
    W = int(rng.choice([64, 96, 160, 200, 320])); H = int(rng.choice([48, 72, 120, 150, 240]))
    seed = int(rng.integers(0, 1 << 30))
    pr = synth.make_pair(seed, W, H, noise=bool(rng.integers(0, 2)), holes=bool(rng.integers(0, 2)))
    c = synth.backproject_numpy(pr.depth_src, pr.intr).reshape(-1, 4).copy()
    if rng.integers(0, 3) == 0:
        c[rng.random(c.shape[0]) < 0.6] = np.nan
    c[:, 3] = rng.integers(0, 2 ** 32, c.shape[0], dtype=np.uint64).astype(np.uint32).view(np.float32)
    thr = float(rng.choice([0.005, 0.02, 0.08, 0.3])); pct = float(rng.choice([0.0, 0.2, 0.5])); mp = int(rng.integers(1, 7)); hyp = int(rng.integers(1, 65))
    leaf = float(rng.choice([0.01, 0.03, 0.1, 0.5]))
    po, lo = O.segment_planes(c, distance_threshold=thr, plane_percent=pct, max_planes=mp, hypotheses=hyp, seed=seed)
    vo = O.voxel_grid(c, leaf, 7.0)
    T = synth.pose_from_seed(seed, max_angle_deg=40.0, max_trans=3.0)
    to, _ = O.pass_transform(c, T, 5.0)
    v2 = O.voxel_grid_only(to, leaf)
    with capi.IcpHandle(capi.default_params(pr.intr, max_batch=1, estimator=capi.EST_SVD)) as h:      # (no ICP here; the svd estimator has no window-moment range limit: 64 x 240 frames have a 150 degree camera)
        pg, lg = h.segment_planes(c.reshape(H, W, 4), h.seg_params(distance_threshold=thr, plane_percent=pct, max_planes=mp, hypotheses=hyp, seed=seed))
        vg = h.voxel_grid(c, leaf)
        tg, _ = h.pass_transform(c, T, 5.0)
        v2g = h.voxel_grid_only(tg, leaf)
    ok = (np.array_equal(lg, lo) and len(pg) == len(po) and all(np.array_equal(a["coeff"], b["coeff"]) and a["count"] == b["count"] for a, b in zip(pg, po))
          and vg.shape == vo.shape and np.array_equal(vg.view(np.uint32), vo.view(np.uint32))
          and np.array_equal(tg.view(np.uint32), to.view(np.uint32)) and v2g.shape == v2.shape and np.array_equal(v2g.view(np.uint32), v2.view(np.uint32)))
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, W=W, H=H, seed=seed, thr=thr, pct=pct, mp=mp, hyp=hyp, leaf=leaf), flush=True)
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)

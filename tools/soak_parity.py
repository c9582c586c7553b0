#!/usr/bin/env python3
"""Randomised parity soak (developer tool, needs an MI355X): random sizes / seeds / gates / estimators / initial guesses,
HIP (default tile-pruned mode) vs the brute-force oracle: indices, d2, every iterate T and the sums must be identical.
usage: tools/soak_parity.py [n_cases] [seed] [nn_mode: 0 auto(tiles) 1 brute VALU 2 brute MFMA]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from slam3d_gx_amd import capi, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
nn_mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = 0
refused = 0
t0 = time.time()
for case in range(n_cases):
    W = int(rng.choice([64, 96, 104, 128, 160, 200, 256, 320]))
    H = int(rng.choice([48, 56, 72, 96, 120, 150, 200, 240]))
    if os.environ.get("SOAK_FULL"):      # full frames (BASELINE's size): long runs only, against the kd-tree oracle
        W, H = 640, 480
    seed = int(rng.integers(0, 1 << 30))
    est = int(rng.integers(0, 3))      # 2: SLAM3D_EST_PLANE (round 5), with and without the pair gate / planes only
    # short runs against the brute-force oracle; long ones (clearance certificates work from the seventh iteration on) against its kd-tree
    iters = int(rng.integers(1, 7)) if rng.random() < 0.4 else int(rng.integers(8, 36))
    if os.environ.get("SOAK_FULL"):
        iters = int(rng.integers(8, 28))
    gate = float(rng.choice([0.01, 0.03, 0.1, 0.3, 1.0]))
    bmd = rng.random() < 0.3           # BASELINE.md section 4's noise and holes (round 5's headline workload), scaled to the frame
    pr = (synth.make_pair(seed, W, H, noise_sigma=0.0012, hole_block=8, hole_prob=0.25) if bmd
          else synth.make_pair(seed, W, H, noise=bool(rng.integers(0, 2)), holes=bool(rng.integers(0, 2))))
    s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
    mode = rng.integers(0, 4)
    Ti = None
    if mode == 1:      # bad initial guess
        Ti = synth.pose_from_seed(seed + 1, max_angle_deg=8.0, max_trans=0.3)
    elif mode == 2:    # sparse target / source
        k = rng.random(s4.shape[:2]) < 0.7; s4 = s4.copy(); s4[k] = np.nan
    elif mode == 3:
        k = rng.random(t4.shape[:2]) < 0.9; t4 = t4.copy(); t4[k] = np.nan
    # half of the cases enter as DEPTH images (the library back-projects them: the target is then camera-consistent and the
    # projective window search takes part), the other half as clouds (tile search alone)
    as_depth = bool(rng.integers(0, 2))
    if as_depth:
        ds, dt = pr.depth_src.copy(), pr.depth_tgt.copy()
        if mode == 2: ds[rng.random(ds.shape) < 0.7] = 0
        if mode == 3: dt[rng.random(dt.shape) < 0.9] = 0
    kw = dict(estimator=est, iterations=iters, max_corr_dist=gate)
    g = int(rng.integers(0, 6))      # one case in three also runs the optional correspondence gates (spec S4g)
    if g == 0:
        kw.update(max_plane_residual2=float(rng.choice([4e-6, 2.5e-5, 1e-4])), min_normal_cos=float(rng.choice([0.0, 0.9, 0.97])))
    elif g == 1:
        kw.update(min_normal_cos=float(rng.choice([0.8, 0.94, 0.985])))
    okw = dict(kw)
    if est == 2:
        fl = int(rng.choice([0, 1, 1, 2, 3]))
        kw.update(plane_flags=fl); okw.update(plane_pair_gate=fl & 1, plane_only=(fl >> 1) & 1)
    po = O.params(pr.intr, nn_method=0 if iters <= 6 else 1, **okw)
    if as_depth:
        s4, t4 = O.backproject(ds, po), O.backproject(dt, po)
    try:
        h = capi.IcpHandle(capi.default_params(pr.intr, max_batch=1, nn_mode=nn_mode, **kw))
    except capi.Slam3dError as e:       # a configuration the library refuses (64 x 240: a 150 degree camera exceeds the window-moment range)
        refused += 1
        continue
    ro = O.icp(s4, t4, po, T_init=Ti)
    with h:
        rg = h.align_depth_batch([ds], [dt], None if Ti is None else [Ti])[0] if as_depth else h.align(s4, t4, Ti)
        idx, d2 = h.get_correspondences(0)
        Tt, St = h.get_trace(0)
    ok = (np.array_equal(idx, ro["idx"]) and np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32))
          and np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:iters], ro["sums_trace"])
          and rg["status"] == ro["status"] and rg["inliers"] == ro["inliers"])
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, W=W, H=H, seed=seed, est=est, iters=iters, gate=gate, mode=int(mode), as_depth=as_depth, kw=kw),
              "idx", int((idx != ro["idx"]).sum()), flush=True)
print(f"{n_cases} cases ({refused} refused at create), {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)

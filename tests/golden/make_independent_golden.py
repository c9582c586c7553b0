#!/usr/bin/env python3
"""Writes tests/golden/independent_golden.json from the numpy / scipy restatement ALONE (no oracle, no HIP):
a complete ICP loop -- scipy.spatial.cKDTree candidates + canonical float32 d^2 re-evaluation for the indices,
numpy.linalg.lstsq on explicit [p x n, n] rows (or numpy.linalg.svd for Kabsch) for the update -- on seeded synthetic
pairs.  Round 4: the target NORMALS too are the restatement's own (whole-frame numpy.linalg.eigh, normals_numpy_full:
spec S2 became well defined with the dominance test, tests/test_oracle_independent.py::test_whole_frame_normals_vs_numpy_eigh),
so nothing of oracle/ is in the chain that produces these numbers.
tests/test_oracle_independent.py then requires oracle/icp_oracle.c to reproduce these index hashes and poses, and
tests/test_golden.py requires the HIP path to reproduce them directly from the depth images: HIP == scipy, oracle == scipy.

usage: python tests/golden/make_independent_golden.py      (CPU only, a few minutes)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import test_oracle_independent as R                      # noqa: E402  (the restatement)
from slam3d_gx_amd import synth                          # noqa: E402

# (seed, width, height, estimator, iterations); seed -1 = the reference's Kinect pair dep/1 -> dep/2 (tests/golden/kinect).
# Round 3: the FULL BASELINE config-2 loop -- 640x480, 20 iterations -- for seeds 1000..1003, the real pair
# for 20 iterations, and one iterate at config 5's 1280x960, all by the scipy restatement alone.
CASES = [(1000, 160, 120, 0, 10), (1001, 160, 120, 1, 10), (1002, 320, 240, 0, 10), (1003, 320, 240, 1, 8),
         (1000, 640, 480, 0, 20), (1001, 640, 480, 0, 20), (1002, 640, 480, 0, 20), (1003, 640, 480, 0, 20),
         (-1, 640, 480, 0, 20), (2000, 1280, 960, 0, 1),
         # round 4: BASELINE.md section 4's own workload (sigma = 0.0012 z^2, 8x8-pixel holes at p = 0.25) and the svd estimator at full size
         (1000, 640, 480, 0, 20, "baseline_md"), (1001, 640, 480, 1, 20),
         # round 5: SLAM3D_EST_PLANE (spec S2p / S4p; estimator 2, a sixth entry = plane_flags: 1 = the plane-pair gate) -- the restatement's
         # own segmentation (python-integer draws, numpy.linalg.eigh refinement), plane normals, association and gate
         (1000, 320, 240, 2, 10, None, 1), (1000, 640, 480, 2, 20, "baseline_md", 1), (1001, 640, 480, 2, 20, "baseline_md", 0),
         (1002, 640, 480, 2, 20, None, 1), (-1, 640, 480, 2, 20, None, 1)]


def icp_numpy(s4, t4, intr, estimator, iterations, gate=0.10, plane_flags=0):
    nrm = snrm = assoc = None
    if estimator == 0:
        nrm = R.normals_numpy_full(t4)[0]                             # the restatement's own S2 (numpy.linalg.eigh)
    elif estimator == 2:                                              # spec S2p: the restatement's own segmentation and plane normals
        nrm, tpl, _ = R.plane_normals_numpy(t4, plane_only=bool(plane_flags & 2))
        if plane_flags & 1:
            snrm, spl, _ = R.plane_normals_numpy(s4, plane_only=bool(plane_flags & 2))
            assoc = R.plane_assoc_numpy(spl, tpl)
    rows = 1 if estimator == 1 else 0
    tgt_ok = R.valid_mask(t4) & ((nrm[..., 3] > 0.5) if nrm is not None else True)
    T = np.eye(4)
    idx = None
    for k in range(iterations):
        idx, ps, sv = R.nn_scipy(s4, t4, tgt_ok, T, gate, coarse=R.is_coarse(k, iterations))       # spec S4c, default three coarse iterations
        if assoc is not None:
            idx = R.pair_gate_numpy(idx, snrm, nrm, assoc)                                          # spec S4p
        T = R.update_from_rows(R.row_vectors(ps, sv, idx, t4, nrm, rows, gate), rows, gate, T)      # spec S4: quantised row vectors
    return idx, T


def main():
    path = os.path.join(HERE, "independent_golden.json")
    out = {"_comment": "written by tests/golden/make_independent_golden.py from the numpy/scipy restatement alone", "cases": []}
    have = {}
    if "--missing" in sys.argv and os.path.exists(path):         # keep what is there, add the cases that are not (same code, same numbers)
        for c in json.load(open(path))["cases"]:
            have[(c["seed"], c["width"], c["height"], c["estimator"], c["iterations"], c.get("workload"), c.get("plane_flags", 0))] = c
    for case in CASES:
        seed, w, h, est, iters = case[:5]
        workload = case[5] if len(case) > 5 else None
        flags = case[6] if len(case) > 6 else 0
        if (seed, w, h, est, iters, workload, flags) in have:
            out["cases"].append(have[(seed, w, h, est, iters, workload, flags)])
            continue
        pr, s4, t4 = R._case(seed, w, h, workload)
        idx, T = icp_numpy(s4, t4, pr.intr, est, iters, plane_flags=flags)
        rot = np.arccos(np.clip((np.trace(np.linalg.inv(pr.T_gt)[:3, :3] @ T[:3, :3]) - 1) / 2, -1, 1)) if seed >= 0 else None
        out["cases"].append(dict(seed=seed, width=w, height=h, estimator=est, iterations=iters, **({"workload": workload} if workload else {}), **({"plane_flags": flags} if est == 2 else {}),
                                 idx_sha256=hashlib.sha256(idx.astype("<i4").tobytes()).hexdigest(), inliers=int((idx >= 0).sum()),
                                 T_final=T.tolist(), rot_err_vs_gt=None if rot is None else float(rot)))
        print(seed, w, h, est, int((idx >= 0).sum()), rot)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Writes tests/golden/independent_golden.json from the numpy / scipy restatement ALONE (no oracle, no HIP):
a complete ICP loop -- scipy.spatial.cKDTree candidates + canonical float32 d^2 re-evaluation for the indices,
numpy.linalg.lstsq on explicit [p x n, n] rows (or numpy.linalg.svd for Kabsch) for the update -- on seeded synthetic
pairs.  Only the target normals are taken from the oracle's S2 (checked against numpy.linalg.eigh in
tests/test_oracle_independent.py::test_normals_vs_numpy_eigh); everything of the iteration itself is independent.
tests/test_oracle_independent.py then requires oracle/icp_oracle.c to reproduce these index hashes and poses, and
tests/test_golden.py requires the HIP path to reproduce the oracle: HIP == oracle == scipy.

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

import oracle_lib as O                                   # noqa: E402  (normals only)
import test_oracle_independent as R                      # noqa: E402  (the restatement)
from slam3d_gx_amd import synth                          # noqa: E402

# (seed, width, height, estimator, iterations); seed -1 = the reference's Kinect pair dep/1 -> dep/2 (tests/golden/kinect).
# Round 3: the FULL BASELINE config-2 loop -- 640x480, 20 iterations -- for seeds 1000..1003, the real pair
# for 20 iterations, and one iterate at config 5's 1280x960, all by the scipy restatement alone.
CASES = [(1000, 160, 120, 0, 10), (1001, 160, 120, 1, 10), (1002, 320, 240, 0, 10), (1003, 320, 240, 1, 8),
         (1000, 640, 480, 0, 20), (1001, 640, 480, 0, 20), (1002, 640, 480, 0, 20), (1003, 640, 480, 0, 20),
         (-1, 640, 480, 0, 20), (2000, 1280, 960, 0, 1)]


def icp_numpy(s4, t4, intr, estimator, iterations, gate=0.10):
    nrm = O.normals(t4, O.params(intr)) if estimator == 0 else None
    tgt_ok = R.valid_mask(t4) & ((nrm[..., 3] > 0.5) if estimator == 0 else True)
    T = np.eye(4)
    idx = None
    for _ in range(iterations):
        idx, ps, sv = R.nn_scipy(s4, t4, tgt_ok, T, gate)
        if estimator == 0:
            A, b, _, _ = R.rows_point2plane(ps, sv, idx, t4, nrm)
            T = R.delta_point2plane(np.linalg.lstsq(A, b, rcond=None)[0]) @ T
        else:
            m = idx[sv] >= 0
            T = R.kabsch(ps[m].astype(np.float64), t4.reshape(-1, 4)[idx[sv][m], :3].astype(np.float64)) @ T
    return idx, T


def main():
    out = {"_comment": "written by tests/golden/make_independent_golden.py from the numpy/scipy restatement alone", "cases": []}
    for seed, w, h, est, iters in CASES:
        pr, s4, t4 = R._case(seed, w, h)
        idx, T = icp_numpy(s4, t4, pr.intr, est, iters)
        rot = np.arccos(np.clip((np.trace(np.linalg.inv(pr.T_gt)[:3, :3] @ T[:3, :3]) - 1) / 2, -1, 1)) if seed >= 0 else None
        out["cases"].append(dict(seed=seed, width=w, height=h, estimator=est, iterations=iters,
                                 idx_sha256=hashlib.sha256(idx.astype("<i4").tobytes()).hexdigest(), inliers=int((idx >= 0).sum()),
                                 T_final=T.tolist(), rot_err_vs_gt=None if rot is None else float(rot)))
        print(seed, w, h, est, int((idx >= 0).sum()), rot)
    json.dump(out, open(os.path.join(HERE, "independent_golden.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

"""Independent restatement of the ICP iteration in numpy / scipy, checked against oracle/icp_oracle.c.

The reference holds no expected pose or index for this path (parity unpinned, PCL cannot be built
here), so the oracle cannot be pinned by the reference.  What CAN be done is to make sure the oracle is not merely
self-consistent: everything below is written against the SPEC with different algorithms and
different code than the oracle uses --

  S2  normals        numpy.linalg.eigh of the window covariance          vs the oracle's adjugate power iteration
  S4  NN indices     scipy.spatial.cKDTree candidates (double precision), then the canonical float32 d^2 and the
                     lowest-index tie-break re-evaluated on the candidates   vs the oracle's brute force / own kd-tree
  S4  29 sums        explicit [p x n, n] rows, A^T A by numpy matmul          vs the oracle's fixed-point accumulation
  S5  p2plane solve  numpy.linalg.lstsq on the explicit rows                  vs the oracle's LDL^T on normal equations
  S5  Kabsch         numpy.linalg.svd                                         vs the oracle's one-sided Jacobi SVD
  S5  update         scipy Rotation.from_euler("xyz") composition             vs the oracle's spec_sincos matrices

Indices must agree EXACTLY; floating-point quantities to 1e-9 (they are computed by different algorithms).
tests/golden/make_independent_golden.py runs the same restatement on the golden cases and is what wrote
tests/golden/independent_golden.json (hashes of the scipy-derived index arrays that the oracle must reproduce).
"""
import json
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

import oracle_lib as O
from slam3d_gx_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------ spec pieces in numpy
def f32(x):
    return np.asarray(x, dtype=np.float32)


def fma32(a, b, c):
    """fmaf(a, b, c) for float32 arrays: the product of two floats is exact in double; the sum is rounded to double and
    then to float (a double rounding that differs from the single rounding only on exact 29-bit halfway patterns,
    ~2^-29 per operation -- never on the fixed seeds used here, and a mismatch would fail loudly)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def transform_f32(T, p):
    """spec S4: R, t rounded once to float; p' = (fmaf(R02, z, fmaf(R01, y, R00 * x)) + t0, ...)"""
    R = f32(T[:3, :3]); t = f32(T[:3, 3])
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = fma32(np.full_like(x, R[r, 2]), z, fma32(np.full_like(x, R[r, 1]), y, R[r, 0] * x)) + t[r]
    return out


def canon_d2(p, q):
    """spec S4: dx = qx - px ...; d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx)) in float"""
    d = q - p
    return fma32(d[:, 2], d[:, 2], fma32(d[:, 1], d[:, 1], d[:, 0] * d[:, 0]))


def valid_mask(c4, zmax=7.0):
    x, y, z = c4[..., 0], c4[..., 1], c4[..., 2]
    return np.isfinite(x) & np.isfinite(y) & np.isfinite(z) & (z > 0) & (z <= np.float32(zmax))


def nn_scipy(src4, tgt4, tgt_ok, T, gate, k=12):
    """exact 1-NN of spec S4 from cKDTree candidates; returns idx[N] (original target pixel index or -1)"""
    N = src4.shape[0] * src4.shape[1]
    s = src4.reshape(-1, 4)[:, :3]; t = tgt4.reshape(-1, 4)[:, :3]
    sv = np.flatnonzero(valid_mask(src4).reshape(-1)); tv = np.flatnonzero(tgt_ok.reshape(-1))
    ps = transform_f32(T, s[sv])
    tree = cKDTree(t[tv].astype(np.float64))
    dist, cand = tree.query(ps.astype(np.float64), k=k)
    g2 = np.float32(gate * gate)
    best_d2 = np.full(len(sv), np.inf, dtype=np.float32); best_j = np.full(len(sv), -1, dtype=np.int64)
    for c in range(k):
        ok = np.isfinite(dist[:, c])
        j = np.where(ok, cand[:, c], 0)
        d2 = np.where(ok, canon_d2(ps, t[tv[j]]), np.float32(np.inf)).astype(np.float32)
        pix = tv[j]
        better = (d2 < best_d2) | ((d2 == best_d2) & (pix < best_j) & ok)       # smallest d2, then smallest pixel index
        best_d2 = np.where(better, d2, best_d2); best_j = np.where(better, pix, best_j)
    # the k-th candidate must be clearly farther than the winner, else k was too small to contain every tie
    far = dist[:, -1].astype(np.float64) ** 2
    assert np.all(~np.isfinite(far) | (far > best_d2.astype(np.float64) * (1 + 1e-4) + 1e-12)), "raise k: near-ties beyond the candidate list"
    idx = np.full(N, -1, dtype=np.int32)
    acc = best_d2 <= g2
    idx[sv[acc]] = best_j[acc]
    return idx, ps, sv


def rows_point2plane(ps, sv, idx, tgt4, nrm4):
    """explicit rows a = [p' x n, n], b = n . (q - p') in double (spec S4)"""
    m = idx[sv] >= 0
    p = ps[m].astype(np.float64)
    j = idx[sv][m]
    q = tgt4.reshape(-1, 4)[j, :3].astype(np.float64)
    n = nrm4.reshape(-1, 4)[j, :3].astype(np.float64)
    A = np.concatenate([np.cross(p, n), n], axis=1)
    b = np.einsum("ij,ij->i", n, q - p)
    return A, b, p, q


def delta_point2plane(x):
    """dR = Rz(g) Ry(b) Rx(a), dt = x[3:6] (spec S5)"""
    D = np.eye(4)
    D[:3, :3] = Rotation.from_euler("xyz", x[:3]).as_matrix()       # extrinsic x, then y, then z = Rz Ry Rx
    D[:3, 3] = x[3:]
    return D


def kabsch(p, q):
    pm, qm = p.mean(0), q.mean(0)
    H = (p - pm).T @ (q - qm)
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    R = Vt.T @ np.diag([1, 1, d]) @ U.T
    D = np.eye(4)
    D[:3, :3] = R; D[:3, 3] = qm - R @ pm
    return D


def normals_numpy(c4, pix, w=7, min_in=41, in_dist=0.01):
    """spec S2 for the listed pixels: (unit normal toward the camera, planar flag), by numpy eigh"""
    H, W = c4.shape[:2]
    ok = valid_mask(c4)
    r = w // 2
    out = []
    for i in pix:
        v, u = divmod(int(i), W)
        if not ok[v, u]:
            out.append((None, False)); continue
        v0, v1, u0, u1 = max(0, v - r), min(H, v + r + 1), max(0, u - r), min(W, u + r + 1)
        win = c4[v0:v1, u0:u1, :3][ok[v0:v1, u0:u1]].astype(np.float64)
        if len(win) < min_in:
            out.append((None, False)); continue
        d = win - c4[v, u, :3].astype(np.float64)
        C = np.cov(d.T, bias=True)
        evals, evecs = np.linalg.eigh(C)
        n = evecs[:, 0]
        if n @ c4[v, u, :3].astype(np.float64) > 0:
            n = -n
        e = (d - d.mean(0)) @ n
        out.append((n, bool((np.abs(e) <= in_dist).sum() >= min_in), evals))
    return out


# ------------------------------------------------------------------------------------------------ the tests
def _case(seed, w, h):
    """seed >= 0: the synthetic pair; seed -1: the reference's Kinect pair data/exp1/dep/1 -> dep/2 (tests/golden/kinect,
    intrinsics of src/convert2PCD.cpp:19-23 = synth.Intrinsics' defaults)"""
    if seed < 0:
        from PIL import Image
        kin = os.path.join(HERE, "golden", "kinect")
        d1 = np.array(Image.open(os.path.join(kin, "exp1_dep_1.png"))).astype(np.uint16)
        d2 = np.array(Image.open(os.path.join(kin, "exp1_dep_2.png"))).astype(np.uint16)
        pr = synth.FramePair(-1, synth.Intrinsics(), d1, d2, np.eye(4))
    else:
        pr = synth.make_pair(seed, w, h)
    return pr, synth.backproject_numpy(pr.depth_src, pr.intr), synth.backproject_numpy(pr.depth_tgt, pr.intr)


@pytest.mark.parametrize("seed,size", [(1000, (320, 240)), (1001, (320, 240)), (1002, (160, 120))])
def test_normals_vs_numpy_eigh(seed, size):
    pr, s4, t4 = _case(seed, *size)
    p = O.params(pr.intr)
    got = O.normals(t4, p).reshape(-1, 4)
    rng = np.random.default_rng(seed)
    pix = rng.choice(size[0] * size[1], 1500, replace=False)
    ref = normals_numpy(t4, pix)
    checked = planar = 0
    for i, r in zip(pix, ref):
        if r[0] is None:
            assert got[i, 3] == 0.0
            continue
        n, flag, evals = r
        # spec S2 (power iteration on the adjugate, five squarings): error angle <= ~(l0 / l1)^32 -- 1e-6 in the cosine needs
        # l1 / l0 >= 1.25; below that the direction of least variance is barely defined and the pixel is not compared
        if evals[1] < 1.25 * evals[0] or evals[1] - evals[0] < 1e-9 * max(evals[2], 1e-30):
            continue
        checked += 1
        if got[i, 3] > 0.5:
            planar += 1
            assert abs(float(got[i, :3].astype(np.float64) @ n)) > 1 - 1e-6, (i, got[i], n)
            assert float(got[i, :3].astype(np.float64) @ n) > 0         # same orientation (towards the camera)
        # the planar decision may differ only where an inlier sits within rounding of the 0.01 m threshold: the
        # oracle's flag must equal numpy's except for such borderline pixels (none on these seeds)
        assert (got[i, 3] > 0.5) == flag, (i, got[i], flag)
    assert checked > 600 and planar > 200


@pytest.mark.parametrize("seed,size,estimator", [(1000, (320, 240), 0), (1001, (320, 240), 1), (1003, (160, 120), 0)])
def test_every_iteration_against_scipy_restatement(seed, size, estimator):
    """For every iterate T_k of the oracle: the scipy NN gives the SAME indices as the oracle's NN at T_k, the explicit
    rows reproduce the oracle's 29 sums, and lstsq / SVD on those rows reproduces T_{k+1}."""
    pr, s4, t4 = _case(seed, *size)
    iters = 8
    p = O.params(pr.intr, estimator=estimator, iterations=iters, nn_method=1)
    ro = O.icp(s4, t4, p)
    nrm = O.normals(t4, p) if estimator == 0 else None
    tgt_ok = valid_mask(t4) & ((nrm[..., 3] > 0.5) if estimator == 0 else True)
    for k in range(iters):
        Tk = ro["T_trace"][k]
        idx, ps, sv = nn_scipy(s4, t4, tgt_ok, Tk, p.max_corr_dist)
        want, _, _ = O.nn_once(s4, t4, O.params(pr.intr, estimator=estimator, nn_method=0), T=Tk, use_normals=(estimator == 0))
        assert np.array_equal(idx, want), f"iteration {k}: {(idx != want).sum()} indices differ between scipy and the oracle"
        S = ro["sums_trace"][k]
        if estimator == 0:
            A, b, _, _ = rows_point2plane(ps, sv, idx, t4, nrm)
            AtA = A.T @ A
            iu = np.triu_indices(6)
            # the oracle's sums are fixed point (spec S4): every term is rounded to a multiple of 2^-32, so a sum of n
            # terms is within n * 2^-33 of the exact one -- a rigorous bound, not a tolerance picked to pass
            tol = len(b) * 2.0 ** -33 * 1.01 + 1e-12
            assert np.abs(S[:21] - AtA[iu]).max() <= tol
            assert np.abs(S[21:27] - A.T @ b).max() <= tol
            assert S[27] == len(b) and abs(S[28] - b @ b) <= tol
            x = np.linalg.lstsq(A, b, rcond=None)[0]
            T_next = delta_point2plane(x) @ Tk
        else:
            m = idx[sv] >= 0
            pp = ps[m].astype(np.float64); qq = t4.reshape(-1, 4)[idx[sv][m], :3].astype(np.float64)
            tol = len(pp) * 2.0 ** -33 * 1.01 + 1e-9
            assert S[27] == len(pp) and np.abs(S[:3] - pp.sum(0)).max() <= tol and np.abs(S[3:6] - qq.sum(0)).max() <= tol
            assert np.abs(S[6:15] - (pp.T @ qq).reshape(9)).max() <= tol
            T_next = kabsch(pp, qq) @ Tk
        assert np.allclose(T_next, ro["T_trace"][k + 1], rtol=0, atol=1e-9), (k, np.abs(T_next - ro["T_trace"][k + 1]).max())
    # and the last iteration's correspondences the oracle reports are those of T_{iters-1}
    idx_last, _, _ = nn_scipy(s4, t4, tgt_ok, ro["T_trace"][iters - 1], p.max_corr_dist)
    assert np.array_equal(idx_last, ro["idx"])


def test_full_size_nn_against_scipy():
    """640x480 (config 2's pair): the scipy restatement gives the oracle's kd-tree indices at the first and at a
    converged iterate (236 k queries each)."""
    pr, s4, t4 = _case(1000, 640, 480)
    p = O.params(pr.intr, iterations=6, nn_method=1)
    ro = O.icp(s4, t4, p)
    nrm = O.normals(t4, p)
    tgt_ok = valid_mask(t4) & (nrm[..., 3] > 0.5)
    for k in (0, 5):
        idx, _, _ = nn_scipy(s4, t4, tgt_ok, ro["T_trace"][k], p.max_corr_dist)
        want, _, _ = O.nn_once(s4, t4, p, T=ro["T_trace"][k], use_normals=True)
        assert np.array_equal(idx, want)
    assert np.array_equal(idx, ro["idx"])


def test_independent_golden_hashes_are_reproduced_by_the_oracle():
    """tests/golden/independent_golden.json was written by make_independent_golden.py from the SCIPY restatement
    alone; the oracle (kd-tree and brute force) must reproduce those index hashes and poses."""
    import hashlib
    G = json.load(open(os.path.join(HERE, "golden", "independent_golden.json")))
    for c in G["cases"]:
        pr, s4, t4 = _case(c["seed"], c["width"], c["height"])
        for nn in (0, 1) if c["width"] <= 160 else (1,):
            ro = O.icp(s4, t4, O.params(pr.intr, estimator=c["estimator"], iterations=c["iterations"], nn_method=nn))
            assert hashlib.sha256(ro["idx"].astype("<i4").tobytes()).hexdigest() == c["idx_sha256"], c
            # 20 chained lstsq solves against 20 chained LDL^T solves of the fixed-point sums: the pose agrees far below the
            # 1e-4 bar of the metric (1e-8 after <= 10 iterations; the real pair's long chain is looser)
            assert np.allclose(ro["T_trace"][-1], np.array(c["T_final"]), rtol=0, atol=1e-8 if c["iterations"] <= 10 else 1e-7), c
            assert ro["inliers"] == c["inliers"]

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


def coarse_mask(H, W):
    """spec S4c: the source pixels that take part in a coarse iteration -- every fourth 8x8-pixel tile, staggered by rows"""
    v, u = np.mgrid[0:H, 0:W]
    return (((u >> 3) + 2 * (v >> 3)) & 3) == 0


def is_coarse(k, iterations, coarse_iterations=3):
    """spec S4c: iteration k of a run is coarse iff it is among the first coarse_iterations and not the run's last"""
    return k < coarse_iterations and k < iterations - 1


def nn_scipy(src4, tgt4, tgt_ok, T, gate, k=12, coarse=False):
    """exact 1-NN of spec S4 from cKDTree candidates; returns idx[N] (original target pixel index or -1)"""
    N = src4.shape[0] * src4.shape[1]
    s = src4.reshape(-1, 4)[:, :3]; t = tgt4.reshape(-1, 4)[:, :3]
    src_ok = valid_mask(src4) & (coarse_mask(*src4.shape[:2]) if coarse else True)
    sv = np.flatnonzero(src_ok.reshape(-1)); tv = np.flatnonzero(tgt_ok.reshape(-1))
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


def b_exponent(gate):
    """spec S4 (round 4; clamp: round 5): EB = 20 - min(k, 8) with gate = m 2^k, 0.5 <= m < 1"""
    return 20 - min(int(np.frexp(float(gate))[1]), 8)


def row_vectors(ps, sv, idx, tgt4, nrm4, estimator, gate):
    """spec S4 (round 4): the 8-component INTEGER row vector of every correspondence, (n, 8) int64 --
    point-to-plane (rint(a 2^16), rint(n 2^20), rint(b 2^EB), 1), a = p' x n, b = n . (q - p');  svd (rint(p' 2^16), rint(q 2^16), 0, 1)"""
    m = idx[sv] >= 0
    p = ps[m].astype(np.float64)
    j = idx[sv][m]
    q = tgt4.reshape(-1, 4)[j, :3].astype(np.float64)
    V = np.zeros((len(p), 8), dtype=np.int64)
    if estimator == 0:
        n = nrm4.reshape(-1, 4)[j, :3].astype(np.float64)
        a = np.stack([p[:, 1] * n[:, 2] - p[:, 2] * n[:, 1], p[:, 2] * n[:, 0] - p[:, 0] * n[:, 2], p[:, 0] * n[:, 1] - p[:, 1] * n[:, 0]], axis=1)
        d = q - p
        b = (n[:, 0] * d[:, 0] + n[:, 1] * d[:, 1]) + n[:, 2] * d[:, 2]
        V[:, 0:3] = np.rint(a * 65536.0).astype(np.int64)
        V[:, 3:6] = np.rint(n * 1048576.0).astype(np.int64)
        V[:, 6] = np.rint(np.ldexp(b, b_exponent(gate))).astype(np.int64)
    else:
        V[:, 0:3] = np.rint(p * 65536.0).astype(np.int64)
        V[:, 3:6] = np.rint(q * 65536.0).astype(np.int64)
    V[:, 7] = 1
    return V


def gram_sums(V, estimator, gate):
    """the 29 doubles the spec derives from the integer Gram matrix G = V^T V (exact int64 arithmetic)"""
    G = V.T @ V                                     # int64 matmul: exact (every total is below 2^60)
    f = lambda x, e: float(np.ldexp(np.float64(int(x)), -e))
    s = np.zeros(29)
    if estimator == 0:
        e = [16, 16, 16, 20, 20, 20]
        eb = b_exponent(gate)
        k = 0
        for r in range(6):
            for c in range(r, 6):
                s[k] = f(G[r, c], e[r] + e[c]); k += 1
        for r in range(6):
            s[21 + r] = f(G[r, 6], e[r] + eb)
        s[27] = float(G[7, 7]); s[28] = f(G[6, 6], 2 * eb)
    else:
        for i in range(3):
            s[i] = f(G[i, 7], 16); s[3 + i] = f(G[3 + i, 7], 16)
        for r in range(3):
            for c in range(3):
                s[6 + 3 * r + c] = f(G[r, 3 + c], 32)
        s[27] = float(G[7, 7])
        s[28] = f(int(G[0, 0]) + int(G[1, 1]) + int(G[2, 2]) + int(G[3, 3]) + int(G[4, 4]) + int(G[5, 5]) - 2 * (int(G[0, 3]) + int(G[1, 4]) + int(G[2, 5])), 32)
    return s


def update_from_rows(V, estimator, gate, T):
    """T_next from the quantised rows: numpy.linalg.lstsq on (A_q, b_q) / numpy.linalg.svd Kabsch on the quantised points"""
    Vf = V.astype(np.float64)
    if estimator == 0:
        A = np.concatenate([Vf[:, 0:3] / 65536.0, Vf[:, 3:6] / 1048576.0], axis=1)
        b = np.ldexp(Vf[:, 6], -b_exponent(gate))
        return delta_point2plane(np.linalg.lstsq(A, b, rcond=None)[0]) @ T
    return kabsch(Vf[:, 0:3] / 65536.0, Vf[:, 3:6] / 65536.0) @ T


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
    """spec S2 for the listed pixels: (unit normal toward the camera, planar flag, eigenvalues), by numpy eigh on the integer
    window moments (round 4b)"""
    H, W = c4.shape[:2]
    ok = valid_mask(c4)
    Xq = np.rint(c4[..., :3] * np.float32(65536.0)).astype(np.float32)
    r = w // 2
    out = []
    for i in pix:
        v, u = divmod(int(i), W)
        if not ok[v, u]:
            out.append((None, False)); continue
        v0, v1, u0, u1 = max(0, v - r), min(H, v + r + 1), max(0, u - r), min(W, u + r + 1)
        win = Xq[v0:v1, u0:u1][ok[v0:v1, u0:u1]].astype(np.int64)
        n = len(win)
        if n < min_in:
            out.append((None, False)); continue
        S1 = win.sum(0); S2 = win.T @ win
        C = (n * S2 - np.outer(S1, S1)).astype(np.float64)
        evals, evecs = np.linalg.eigh(C)
        nvec = evecs[:, 0]
        if nvec @ Xq[v, u].astype(np.float64) > 0:
            nvec = -nvec
        nf = nvec.astype(np.float32)
        dqf = np.float32((nvec[0] * (S1[0] * (1.0 / n)) + nvec[1] * (S1[1] * (1.0 / n))) + nvec[2] * (S1[2] * (1.0 / n)))
        wf = win.astype(np.float32)
        e = fma32(np.full(n, nf[2]), wf[:, 2], fma32(np.full(n, nf[1]), wf[:, 1], nf[0] * wf[:, 0])) - dqf
        out.append((nvec, bool((np.abs(e) <= np.float32(in_dist * 65536.0)).sum() >= min_in), evals))
    return out


def normals_numpy_full(c4, w=7, min_in=41, in_dist=0.01, band=32, zmax=7.0):
    """spec S2 (round 4b: integer window moments) for a WHOLE frame, vectorised numpy + numpy.linalg.eigh -- no oracle code:
    (H, W, 4) float32 normals with the planar flag in .w, and the eigenvalue ratio l1 / l0 of every window that had enough
    points.  Differences of algorithm against oracle/icp_oracle.c: int64 einsum moments in one go (the oracle: a loop), eigh of
    C' = n S2 - S1 S1^T (the oracle: seven squarings of the adjugate), the dominance rule from the eigenvalues (the oracle:
    ||M||_F^2 against tr(M)^2 of the squared adjugate -- identical in exact arithmetic: 1 - ||M||_F^2 / tr(M)^2 =
    2 (r1 + r2 + r1 r2) / (1 + r1 + r2)^2 with r1 = (l0 / l1)^128, r2 = (l0 / l2)^128)."""
    from numpy.lib.stride_tricks import sliding_window_view
    H, W = c4.shape[:2]
    r = w // 2
    ok = valid_mask(c4, zmax)
    Xq = np.where(ok[..., None], np.rint(c4[..., :3] * np.float32(65536.0)), np.float32(0)).astype(np.float32)      # rintf(x * 2^16): integers below 2^20
    P = np.zeros((H + 2 * r, W + 2 * r, 3), dtype=np.int64)
    M = np.zeros((H + 2 * r, W + 2 * r), dtype=bool)
    P[r:r + H, r:r + W] = Xq.astype(np.int64)
    M[r:r + H, r:r + W] = ok
    out = np.zeros((H, W, 4), dtype=np.float32)
    ratio = np.full((H, W), np.inf)
    thr = np.float32(in_dist * 65536.0)
    for v0 in range(0, H, band):
        v1 = min(H, v0 + band)
        Pw = sliding_window_view(P[v0:v1 + 2 * r], (w, w), axis=(0, 1))      # (b, W, 3, w, w) int64, zeros where invalid
        Mw = sliding_window_view(M[v0:v1 + 2 * r], (w, w), axis=(0, 1))      # (b, W, w, w)
        c0 = P[v0 + r:v1 + r, r:r + W].astype(np.float64)
        n = Mw.sum(axis=(2, 3)).astype(np.int64)
        cen = ok[v0:v1] & (n >= min_in)
        S1 = Pw.sum(axis=(3, 4))                                               # exact integers
        S2 = np.einsum("bwixy,bwjxy->bwij", Pw, Pw)
        C = (n[..., None, None] * S2 - S1[..., :, None] * S1[..., None, :]).astype(np.float64)      # exact: every entry below 2^53
        evals, evecs = np.linalg.eigh(C)
        l0, l1, l2 = np.abs(evals[..., 0]), evals[..., 1], evals[..., 2]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
            r1 = np.where(l1 > 0, (l0 / l1) ** 128, np.inf)
            r2 = np.where(l2 > 0, (l0 / l2) ** 128, np.inf)
            defect = 2.0 * (r1 + r2 + r1 * r2) / (1.0 + r1 + r2) ** 2        # 1 - ||M||_F^2 / tr(M)^2 of the squared adjugate
            dominant = defect <= 2.0 ** -40
            ratio[v0:v1] = np.where(cen, l1 / np.maximum(l0, 1e-300), np.inf)
        nv = evecs[..., 0]
        flip = (nv * c0).sum(-1) > 0
        nv = np.where(flip[..., None], -nv, nv)
        nn = np.where(n > 0, n, 1).astype(np.float64)
        mean = S1.astype(np.float64) * (1.0 / nn)[..., None]
        dqf = ((nv[..., 0] * mean[..., 0] + nv[..., 1] * mean[..., 1]) + nv[..., 2] * mean[..., 2]).astype(np.float32)
        nf = nv.astype(np.float32)
        Pf = Pw.astype(np.float32)                                              # the quantised coordinates as floats (exact)
        bc = lambda a: np.broadcast_to(a[..., None, None], Pf[:, :, 0].shape)
        e = fma32(bc(nf[..., 2]), Pf[:, :, 2], fma32(bc(nf[..., 1]), Pf[:, :, 1], bc(nf[..., 0]) * Pf[:, :, 0])) - bc(dqf)
        cnt = ((np.abs(e) <= thr) & Mw).sum(axis=(2, 3))
        good = cen & dominant & (cnt >= min_in)
        out[v0:v1, :, :3] = np.where(good[..., None], nf, np.float32(0))
        out[v0:v1, :, 3] = good
    return out, ratio


# ------------------------------------------------------------------------------------------------ spec P1-P5 / S2p / S4p in numpy
_GOLD = 0x9E3779B97F4A7C15
_M64 = (1 << 64) - 1


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def segment_planes_numpy(c4, zmax=7.0, thr=0.04, percent=0.2, max_planes=3, hypotheses=64, seed=1, draws=32):
    """Spec P1-P5 (the pcl::SACSegmentation loop of src/GraphicEnd.cpp:353-430) written against the SPEC
    with numpy: python integers for the counter-based draws, vectorised float32 consensus tests, int64 moments,
    numpy.linalg.eigh for the refinement (the oracle: cyclic Jacobi).  -> (planes [n, 8] float32: a b c d cx cy cz count, labels [N])"""
    P = c4.reshape(-1, 4)[:, :3].astype(np.float32)
    n = P.shape[0]
    ok = valid_mask(c4.reshape(-1, 4), zmax)
    lab = np.where(ok, -1, -2).astype(np.int32)
    n_valid = int(ok.sum())
    remaining = n_valid
    thr = np.float32(thr); percent = np.float32(percent)
    planes = []
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    for r in range(max_planes):
        if n_valid < 3 or not (float(remaining) > float(percent) * float(n_valid)):
            break
        H = []
        for h in range(hypotheses):
            st = (seed + _GOLD * (1 + r * 4096 + h)) & _M64
            pick = []
            for _ in range(draws):
                if len(pick) == 3:
                    break
                st = (st + _GOLD) & _M64
                pix = ((_mix64(st) >> 32) * n) >> 32
                if lab[pix] != -1 or pix in pick[:2]:
                    continue
                pick.append(pix)
            if len(pick) < 3:
                H.append(None); continue
            p0, p1, p2 = P[pick[0]], P[pick[1]], P[pick[2]]
            a = p1 - p0; b = p2 - p0
            one = lambda v: np.array([v], dtype=np.float32)
            nx = fma32(one(a[1]), one(b[2]), -one(a[2] * b[1]))[0]
            ny = fma32(one(a[2]), one(b[0]), -one(a[0] * b[2]))[0]
            nz = fma32(one(a[0]), one(b[1]), -one(a[1] * b[0]))[0]
            nn = fma32(one(nz), one(nz), fma32(one(ny), one(ny), one(nx * nx)))[0]
            if not nn > np.float32(1e-16):
                H.append(None); continue
            dd = -fma32(one(nx), one(p0[0]), fma32(one(ny), one(p0[1]), one(nz * p0[2])))[0]
            H.append((nx, ny, nz, dd, np.float32(thr * thr) * nn, p0))
        un = lab == -1
        xs, ys, zs = x[un], y[un], z[un]
        def inl(hy, xs=xs, ys=ys, zs=zs):
            nx, ny, nz, dd, t2, _ = hy
            e = fma32(np.full_like(xs, nx), xs, fma32(np.full_like(xs, ny), ys, nz * zs)) + dd
            return e * e <= t2
        cnt = [int(inl(hy).sum()) if hy is not None else 0 for hy in H]
        best = int(np.argmax(cnt))                      # first maximum: ties -> lowest h
        if cnt[best] < 3:
            break
        hy = H[best]
        m = inl(hy)
        o = hy[5].astype(np.float64)
        q = np.stack([np.rint((xs[m].astype(np.float64) - o[0]) * 65536.0), np.rint((ys[m].astype(np.float64) - o[1]) * 65536.0),
                      np.rint((zs[m].astype(np.float64) - o[2]) * 65536.0)], axis=1).astype(np.int64)
        S0 = len(q); S1 = q.sum(0); S2 = q.T @ q
        inv = 1.0 / S0
        mean = S1.astype(np.float64) * inv
        C = S2.astype(np.float64) * inv - np.outer(mean, mean)
        evals, evecs = np.linalg.eigh(C)
        nv = evecs[:, 0] / np.linalg.norm(evecs[:, 0])
        c = o + mean / 65536.0
        d = -((nv[0] * c[0] + nv[1] * c[1]) + nv[2] * c[2])
        if d < 0:
            nv, d = -nv, -d
        af, bf, cf, df = np.float32(nv[0]), np.float32(nv[1]), np.float32(nv[2]), np.float32(d)
        e = fma32(np.full_like(xs, af), xs, fma32(np.full_like(xs, bf), ys, cf * zs)) + df
        on = np.abs(e) <= thr
        got = int(on.sum())
        if got == 0:
            break
        idx_un = np.flatnonzero(un)
        lab[idx_un[on]] = r
        planes.append([af, bf, cf, df, np.float32(c[0]), np.float32(c[1]), np.float32(c[2]), np.float32(got)])
        remaining -= got
    return np.array(planes, dtype=np.float32).reshape(-1, 8), lab


def plane_normals_numpy(c4, plane_only=False, seg=None, **kw):
    """Spec S2p: a pixel labelled with plane r takes (a, b, c) of the plane, w = 1 + r; a pixel on no plane keeps its S2 window
    normal with w = 0.75 (plane_only: nothing).  No oracle code: segment_planes_numpy + normals_numpy_full."""
    planes, lab = segment_planes_numpy(c4, **(seg or {}))
    H, W = c4.shape[:2]
    out = np.zeros((H * W, 4), dtype=np.float32)
    if not plane_only:
        win = normals_numpy_full(c4, **kw)[0].reshape(-1, 4)
        m = win[:, 3] > 0.5
        out[m, :3] = win[m, :3]; out[m, 3] = 0.75
    for r in range(len(planes)):
        m = lab == r
        out[m, :3] = planes[r, :3]; out[m, 3] = 1 + r
    return out.reshape(H, W, 4), planes, lab


def plane_assoc_numpy(P1, P2, T=None):
    """Spec S4p: planes of frame 1 carried by T (n' = R n, d' = d - n'.t, sign d' >= 0), nearest plane of frame 2 on (a, b, c, d)"""
    T = np.eye(4) if T is None else np.asarray(T, dtype=np.float64)
    out = np.full(len(P1), -1, dtype=np.int32)
    for i in range(len(P1)):
        a, b, c, d = [float(v) for v in P1[i, :4]]
        n = [(T[r, 0] * a + T[r, 1] * b) + T[r, 2] * c for r in range(3)]
        dd = d - ((n[0] * T[0, 3] + n[1] * T[1, 3]) + n[2] * T[2, 3])
        if dd < 0:
            n, dd = [-v for v in n], -dd
        m = np.array([*n, dd]).astype(np.float32)
        best, bd = -1, np.float32(np.inf)
        for j in range(len(P2)):
            d2 = np.zeros(1, dtype=np.float32)
            for k in range(4):
                e = np.array([m[k] - P2[j, k]], dtype=np.float32)
                d2 = fma32(e, e, d2)
            if d2[0] < bd:
                bd, best = d2[0], j
        out[i] = best
    return out


def pair_gate_numpy(idx, snrm, tnrm, assoc):
    """Spec S4p: a correspondence is kept iff the target pixel's plane is the one associated with the source pixel's plane
    (label = int(normal.w) - 1; a pixel on no plane: -1, associated with -1)"""
    ls = snrm.reshape(-1, 4)[:, 3].astype(np.int32) - 1
    lt = tnrm.reshape(-1, 4)[:, 3].astype(np.int32) - 1
    has = idx >= 0
    want = np.where(ls >= 0, np.concatenate([assoc, [-1]])[np.where(ls >= 0, np.minimum(ls, len(assoc)), len(assoc))] if len(assoc) else -1, -1)
    keep = has & (want == lt[np.where(has, idx, 0)])
    return np.where(keep, idx, -1).astype(np.int32)


# ------------------------------------------------------------------------------------------------ the tests
BASELINE_MD = dict(noise_sigma=0.0012, hole_block=8, hole_prob=0.25)      # BASELINE.md section 4's synthetic workload


def _case(seed, w, h, workload=None):
    """seed >= 0: the synthetic pair (workload "baseline_md": BASELINE.md section 4's noise and hole mask instead of the
    defaults); seed -1: the reference's Kinect pair data/exp1/dep/1 -> dep/2 (tests/golden/kinect, intrinsics of
    src/convert2PCD.cpp:19-23 = synth.Intrinsics' defaults)"""
    if seed < 0:
        from PIL import Image
        kin = os.path.join(HERE, "golden", "kinect")
        d1 = np.array(Image.open(os.path.join(kin, "exp1_dep_1.png"))).astype(np.uint16)
        d2 = np.array(Image.open(os.path.join(kin, "exp1_dep_2.png"))).astype(np.uint16)
        pr = synth.FramePair(-1, synth.Intrinsics(), d1, d2, np.eye(4))
    else:
        pr = synth.make_pair(seed, w, h, **(BASELINE_MD if workload == "baseline_md" else {}))
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


@pytest.mark.parametrize("seed,size,kw", [(1000, (640, 480), {}), (1001, (320, 240), {}),
                                          (1000, (640, 480), dict(noise_sigma=0.0012, hole_block=8, hole_prob=0.25)), (-2, (640, 480), {})])
def test_whole_frame_normals_vs_numpy_eigh(seed, size, kw):
    """Round 4: with seven squarings and the dominance test, spec S2 is well defined -- a
    window either has a direction of least variance to ~1e-12 or no normal at all -- so numpy.linalg.eigh must reproduce
    the oracle's planar flags on EVERY pixel of a frame and the float normals bit for bit (a last-bit difference needs the
    double to sit within ~1e-12 of a float rounding boundary: allowed on a handful of components, none seen on these
    frames).  Cases: the headline workload, BASELINE.md's (sigma = 0.0012 z^2 + 8x8 holes, where 4 % of the flagged
    windows have l1 / l0 below 2), and the reference's Kinect frame bin/dep_1.png (seed -2)."""
    if seed == -2:
        from PIL import Image
        d = np.array(Image.open(os.path.join(HERE, "golden", "kinect", "bin_dep_1.png"))).astype(np.uint16)
        intr = synth.Intrinsics()
        t4 = synth.backproject_numpy(d, intr)
    else:
        pr = synth.make_pair(seed, *size, **kw)
        intr = pr.intr
        t4 = synth.backproject_numpy(pr.depth_tgt, intr)
    got = O.normals(t4, O.params(intr))
    ref, ratio = normals_numpy_full(t4)
    fo, fn = got[..., 3] > 0.5, ref[..., 3] > 0.5
    assert fo.sum() > 0.1 * fo.size * (0.5 if kw else 1.0)
    assert np.array_equal(fo, fn), f"{(fo != fn).sum()} planar flags differ"
    assert ratio[fo].min() > 1.2                                   # what carries a normal has a well-defined direction
    # (x + 0.0 turns -0.0 into +0.0: an exactly fronto-parallel window -- constant quantised depth -- gives (+-0, +-0, -1), and the
    # sign of a zero changes no product, sum or comparison downstream)
    diff = ((got[..., :3] + np.float32(0)).view(np.uint32) != (ref[..., :3] + np.float32(0)).view(np.uint32)) & fo[..., None]
    assert diff.sum() <= 3, f"{diff.sum()} float components differ"
    dots = (got[..., :3].astype(np.float64) * ref[..., :3].astype(np.float64)).sum(-1)[fo]
    assert dots.min() > 1 - 1e-6


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
        cz = is_coarse(k, iters, p.coarse_iterations)
        idx, ps, sv = nn_scipy(s4, t4, tgt_ok, Tk, p.max_corr_dist, coarse=cz)
        want, _, _ = O.nn_once(s4, t4, O.params(pr.intr, estimator=estimator, nn_method=0), T=Tk, use_normals=(estimator == 0), coarse=cz)
        assert np.array_equal(idx, want), f"iteration {k}: {(idx != want).sum()} indices differ between scipy and the oracle"
        S = ro["sums_trace"][k]
        V = row_vectors(ps, sv, idx, t4, nrm, estimator, p.max_corr_dist)
        # spec S4 (round 4): the sums are DERIVED from the exact integer Gram matrix of the quantised row vectors, so the numpy
        # restatement (np.rint, int64 matmul, ldexp) must reproduce every one of the oracle's 29 doubles EXACTLY -- no tolerance
        assert np.array_equal(S, gram_sums(V, estimator, p.max_corr_dist)), (k, np.abs(S - gram_sums(V, estimator, p.max_corr_dist)).max())
        if estimator == 0:      # and the quantised sums are the plain double-precision ones to the quantisation bound
            A, bb, _, _ = rows_point2plane(ps, sv, idx, t4, nrm)
            AtA = A.T @ A
            iu = np.triu_indices(6)
            amax = np.abs(A).max()
            assert np.abs(S[:21] - AtA[iu]).max() <= len(bb) * (2.0 ** -17 * 2 * amax + 2.0 ** -34) * 1.01
            assert S[27] == len(bb) and abs(S[28] - bb @ bb) <= len(bb) * 2.0 ** -(b_exponent(p.max_corr_dist)) * 0.11
        T_next = update_from_rows(V, estimator, p.max_corr_dist, Tk)
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
        cz = is_coarse(k, 6, p.coarse_iterations)
        idx, _, _ = nn_scipy(s4, t4, tgt_ok, ro["T_trace"][k], p.max_corr_dist, coarse=cz)
        want, _, _ = O.nn_once(s4, t4, p, T=ro["T_trace"][k], use_normals=True, coarse=cz)
        assert np.array_equal(idx, want)
    assert (idx >= 0).sum() > 3.5 * (nn_scipy(s4, t4, tgt_ok, ro["T_trace"][0], p.max_corr_dist, coarse=True)[0] >= 0).sum()      # a coarse iteration uses a quarter
    assert np.array_equal(idx, ro["idx"])


def test_independent_golden_hashes_are_reproduced_by_the_oracle():
    """tests/golden/independent_golden.json was written by make_independent_golden.py from the SCIPY restatement
    alone; the oracle (kd-tree and brute force) must reproduce those index hashes and poses."""
    import hashlib
    G = json.load(open(os.path.join(HERE, "golden", "independent_golden.json")))
    for c in G["cases"]:
        pr, s4, t4 = _case(c["seed"], c["width"], c["height"], c.get("workload"))
        for nn in (0, 1) if c["width"] <= 160 else (1,):
            fl = c.get("plane_flags", 0)
            ro = O.icp(s4, t4, O.params(pr.intr, estimator=c["estimator"], iterations=c["iterations"], nn_method=nn, plane_pair_gate=fl & 1, plane_only=(fl >> 1) & 1))
            assert hashlib.sha256(ro["idx"].astype("<i4").tobytes()).hexdigest() == c["idx_sha256"], c
            # 20 chained lstsq solves against 20 chained LDL^T solves of the fixed-point sums: the pose agrees far below the
            # 1e-4 bar of the metric (1e-8 after <= 10 iterations; the real pair's long chain is looser)
            assert np.allclose(ro["T_trace"][-1], np.array(c["T_final"]), rtol=0, atol=1e-8 if c["iterations"] <= 10 else 1e-7), c
            assert ro["inliers"] == c["inliers"]


@pytest.mark.parametrize("case", [(1000, (320, 240), {}), (1002, (640, 480), BASELINE_MD), (-2, (640, 480), {}), (-1, (640, 480), {})])
def test_plane_normals_against_the_numpy_restatement(case):
    """Spec S2p with no oracle code in the chain (python-integer draws, vectorised float32 consensus, int64 moments,
    numpy.linalg.eigh refinement, whole-frame eigh window normals): the oracle's planes, labels and per-pixel normals must be
    reproduced exactly -- the headline geometry, BASELINE.md's workload, and the reference's Kinect frames."""
    seed, size, kw = case
    if seed < 0:
        from PIL import Image
        name = "bin_dep_1.png" if seed == -2 else "exp1_dep_2.png"
        intr = synth.Intrinsics()
        t4 = synth.backproject_numpy(np.array(Image.open(os.path.join(HERE, "golden", "kinect", name))).astype(np.uint16), intr)
    else:
        pr = synth.make_pair(seed, *size, **kw)
        intr = pr.intr
        t4 = synth.backproject_numpy(pr.depth_tgt, intr)
    p = O.params(intr, estimator=2)
    got, planes, labels = O.plane_normals(t4, p)
    want, planes_np, labels_np = plane_normals_numpy(t4)
    assert len(planes) == len(planes_np) >= 1
    assert np.array_equal(labels, labels_np), f"{(labels != labels_np).sum()} labels differ"
    assert np.array_equal(planes[:, 7], planes_np[:, 7])
    assert np.abs(planes[:, :7].astype(np.float64) - planes_np[:, :7]).max() < 1e-6
    assert (planes[:, :4].view(np.uint32) != planes_np[:, :4].view(np.uint32)).sum() <= 1      # a last-bit difference needs a double within ~1e-15 of a float rounding boundary
    assert np.array_equal(got[..., 3], want[..., 3])
    assert ((got[..., :3] + np.float32(0)).view(np.uint32) != (want[..., :3] + np.float32(0)).view(np.uint32)).sum() <= 3 * (labels >= 0).sum() * ((planes[:, :3] != planes_np[:, :3]).any()) + 3


@pytest.mark.parametrize("seed,size,gate", [(1000, (320, 240), 1), (1003, (160, 120), 0)])
def test_every_iteration_of_the_plane_estimator_against_the_restatement(seed, size, gate):
    """SLAM3D_EST_PLANE (spec S2p / S4p) at every iterate of the oracle: scipy NN over the targets that carry a (plane or window)
    normal gives the oracle's indices, the plane-pair gate in numpy keeps the same correspondences, the integer Gram sums are
    equal exactly and lstsq reproduces the next pose."""
    pr, s4, t4 = _case(seed, *size)
    iters = 6
    p = O.params(pr.intr, estimator=2, iterations=iters, nn_method=1, plane_pair_gate=gate)
    ro = O.icp(s4, t4, p)
    tn, tpl, _ = plane_normals_numpy(t4)
    sn, spl, _ = plane_normals_numpy(s4)
    assoc = plane_assoc_numpy(spl, tpl)
    assert np.array_equal(assoc, O.plane_assoc(np.asarray(spl), np.asarray(tpl)))
    tgt_ok = valid_mask(t4) & (tn[..., 3] > 0.5)
    for k in range(iters):
        Tk = ro["T_trace"][k]
        cz = is_coarse(k, iters, p.coarse_iterations)
        idx, ps, sv = nn_scipy(s4, t4, tgt_ok, Tk, p.max_corr_dist, coarse=cz)
        want, _, _ = O.nn_once(s4, t4, O.params(pr.intr, estimator=2, nn_method=0), T=Tk, use_normals=2, coarse=cz)
        assert np.array_equal(idx, want), k
        if gate:
            idx = pair_gate_numpy(idx, sn, tn, assoc)
        V = row_vectors(ps, sv, idx, t4, tn, 0, p.max_corr_dist)
        assert np.array_equal(ro["sums_trace"][k], gram_sums(V, 0, p.max_corr_dist)), k
        T_next = update_from_rows(V, 0, p.max_corr_dist, Tk)
        assert np.allclose(T_next, ro["T_trace"][k + 1], rtol=0, atol=1e-9), k
    idx_last = nn_scipy(s4, t4, tgt_ok, ro["T_trace"][iters - 1], p.max_corr_dist)[0]
    if gate:
        idx_last = pair_gate_numpy(idx_last, sn, tn, assoc)
    assert np.array_equal(idx_last, ro["idx"])


def _unquantised_run(s4, t4, intr, T_init=None, iterations=20, gate=0.10):
    """SURVEY.md App. C4 as written: per iteration the exact correspondences at the current pose (the oracle's NN: the indices are
    not what is under test), the rows a = [p' x n, n], b = n . (q - p') in DOUBLE, UNquantised, A^T A and A^T b accumulated in
    double by numpy.linalg.lstsq, dR = Rz Ry Rx.  Returns the pose after `iterations` iterations."""
    p = O.params(intr, iterations=iterations, nn_method=1, max_corr_dist=gate)
    nrm = O.normals(t4, p)
    s = s4.reshape(-1, 4)[:, :3]
    T = np.eye(4) if T_init is None else np.array(T_init, dtype=np.float64)
    step = 0.0
    for it in range(iterations):
        coarse = is_coarse(it, iterations)
        idx, _, _ = O.nn_once(s4, t4, p, T=T, use_normals=True, coarse=coarse)
        sv = np.flatnonzero(valid_mask(s4).reshape(-1))
        ps = transform_f32(T, s[sv])
        A, b, _, _ = rows_point2plane(ps, sv, idx, t4, nrm)
        Tn = delta_point2plane(np.linalg.lstsq(A, b, rcond=None)[0]) @ T
        # the SAME correspondences through the spec's integer rows: what the quantisation alone does to one update
        Tq = update_from_rows(row_vectors(ps, sv, idx, t4, nrm, 0, gate), 0, gate, T)
        step = max(step, *O.pose_error(Tn, Tq))
        T = Tn
    return T, step


@pytest.mark.parametrize("workload", ["low_noise", "baseline_md"])
def test_integer_rows_stay_within_1e6_of_the_double_accumulation(workload):
    """VERDICT r5 item 7a.  SURVEY.md App. C4 says: accumulate A^T A, A^T b in double.  Spec S4 quantises every row to integers first
    (a 2^16, n 2^20, b 2^EB) so that the sums are order-free; the scipy restatement above shares that quantisation, so until now no test
    said how far the quantised pose is from the unquantised, double-accumulated one.  This one does: seeds 1000..1003 under both synthetic
    workloads at 640x480 -- the spec's pose (the oracle's, which the HIP path reproduces bit for bit) against a 20-iteration run whose
    every update comes from unquantised double rows through numpy.linalg.lstsq.  Two bounds: (i) ONE update from the same
    correspondences, integer rows against double rows: <= 1e-6 rad / 1e-6 m (measured <= 8.4e-8) -- what the quantisation itself does; (ii) the poses after
    20 iterations: <= 1e-5 rad / 1e-5 m, a tenth of the metric's bar -- the runs have not all reached a fixed point by then (the pose of
    seed 1002 still moves 1e-5 per iteration), so a 1e-9 difference flips a handful of correspondences on the way and the end poses
    differ by more than the updates do (measured <= 1.6e-6 rad / 8.8e-6 m; the converging Kinect self-alignments below: <= 1e-7 m).
    The measured worst cases are printed; DESIGN.md section 3 quotes them."""
    worst = (0.0, 0.0, 0.0)
    for seed in (1000, 1001, 1002, 1003):
        pr, s4, t4 = _case(seed, 640, 480, workload)
        spec = O.icp(s4, t4, O.params(pr.intr, iterations=20, nn_method=1))["T_trace"][-1]
        Tu, step = _unquantised_run(s4, t4, pr.intr)
        rot, tr = O.pose_error(spec, Tu)
        assert step <= 1e-6, (workload, seed, step)
        assert rot <= 1e-5 and tr <= 1e-5, (workload, seed, rot, tr)
        worst = (max(worst[0], rot), max(worst[1], tr), max(worst[2], step))
    print(f"S4 quantisation vs double accumulation ({workload}): one update <= {worst[2]:.1e} rad|m; after 20 iterations worst {worst[0]:.2e} rad / {worst[1]:.2e} m")


def test_integer_rows_vs_double_accumulation_on_the_reference_kinect_frames():
    """The same bound on the reference's frames: the perturbed self-alignments dep1 -> dep1 and dep2 -> dep2 (2 degrees / 3 cm off)."""
    from PIL import Image
    kin = os.path.join(HERE, "golden", "kinect")
    intr = synth.Intrinsics()
    Ti = synth.pose_from_seed(77, 2.0, 0.03)
    for name in ("exp1_dep_1.png", "exp1_dep_2.png"):
        d = np.array(Image.open(os.path.join(kin, name))).astype(np.uint16)
        c4 = synth.backproject_numpy(d, intr)
        spec = O.icp(c4, c4, O.params(intr, iterations=20, nn_method=1), T_init=Ti)["T_trace"][-1]
        Tu, step = _unquantised_run(c4, c4, intr, T_init=Ti)
        rot, tr = O.pose_error(spec, Tu)
        assert step <= 1e-6 and rot <= 1e-5 and tr <= 1e-5, (name, step, rot, tr)
        print(f"S4 quantisation vs double accumulation ({name} perturbed): one update <= {step:.1e}; after 20 iterations {rot:.2e} rad / {tr:.2e} m")

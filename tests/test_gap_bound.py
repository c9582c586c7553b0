"""The pruning bound of the tile search (slam3d_gx_amd/csrc/icp_kernels.hpp: lane_gap_le / box_gap2).

A target tile (or quadrant, or coarse cell) is skipped when the squared GAP between a query and the tile's axis-aligned
box exceeds the query's current best d^2.  Exactness of the search rests on   gap2(p, box) <= canon_d2(p, q)   for
EVERY point q inside the box, in float arithmetic: each |q_i - p_i| >= gap_i because rounding of a subtraction is
monotone, and the sum has the same fma association as the canonical distance, which is monotone in each non-negative
term.  So the bound holds EXACTLY, with no margin (the kernel still multiplies its threshold by 1.00001, VERDICT r1
asked for this property test).  Checked here on millions of (point, box, q) triples including the box corners, the
closest point of the box and points on its faces, at the coordinate magnitudes of the path (0.5 .. 7 m) and at
degenerate boxes.  Pure numpy emulation of the float32 operations (a float product is exact in double)."""
import numpy as np
import pytest


def f32(x):
    return np.asarray(x, dtype=np.float32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def canon_d2(p, q):
    d = q - p                                                     # float32 subtraction, rounded once
    return fma32(d[..., 2], d[..., 2], fma32(d[..., 1], d[..., 1], d[..., 0] * d[..., 0]))


def gap2(p, lo, hi):
    g = np.maximum(np.float32(0), np.maximum(lo - p, p - hi))     # v_sub, v_max: exactly the kernel's operations
    return fma32(g[..., 2], g[..., 2], fma32(g[..., 1], g[..., 1], g[..., 0] * g[..., 0]))


@pytest.mark.parametrize("seed", range(6))
def test_gap_is_a_lower_bound_of_the_canonical_distance(seed):
    rng = np.random.default_rng(seed)
    n = 400_000
    scale = [7.0, 7.0, 7.0, 0.05, 1e-3, 100.0][seed]               # scene scale, tiny boxes, sub-millimetre, far away
    c = f32(rng.uniform(-scale, scale, (n, 3)))
    ext = f32(rng.uniform(0, [0.3, 0.3, 0.3, 0.01, 1e-4, 1.0][seed], (n, 3)))
    ext[rng.random((n, 3)) < 0.1] = 0                              # degenerate (flat / single-point) boxes
    lo, hi = c - ext, c + ext
    lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)
    p = f32(c + rng.normal(scale=[0.5, 0.1, 0.02, 0.02, 1e-3, 5.0][seed], size=(n, 3)))
    g2 = gap2(p, lo, hi)
    # q: the 8 corners, the closest point of the box, random interior points, random points on the faces
    cands = []
    for m in range(8):
        cands.append(np.where([(m >> k) & 1 for k in range(3)], hi, lo).astype(np.float32))
    cands.append(np.clip(p, lo, hi))
    for _ in range(6):
        t = f32(rng.random((n, 3)))
        q = (lo.astype(np.float64) + t * (hi.astype(np.float64) - lo.astype(np.float64))).astype(np.float32)
        q = np.clip(q, lo, hi)
        face = rng.integers(0, 3, n)
        side = rng.random(n) < 0.5
        qf = q.copy()
        qf[np.arange(n), face] = np.where(side, lo[np.arange(n), face], hi[np.arange(n), face])
        cands += [q, qf]
    worst = np.float32(np.inf)
    for q in cands:
        d2 = canon_d2(p, q)
        assert np.all(g2 <= d2), f"gap bound violated: {int((g2 > d2).sum())} of {n}"
        worst = min(worst, (d2 - g2).min())
    # the bound is tight: at the closest point it is attained (difference exactly zero somewhere)
    assert np.any(canon_d2(p, np.clip(p, lo, hi)) == g2)
    inside = np.all((p >= lo) & (p <= hi), axis=1)
    assert np.all(g2[inside] == 0)


def test_box_to_box_gap_is_a_lower_bound_too():
    """box_gap2 (wave-level test: a tile box against the wave's query box) bounds the gap of every query in the box."""
    rng = np.random.default_rng(42)
    n = 300_000
    qc = f32(rng.uniform(-5, 5, (n, 3))); qe = f32(rng.uniform(0, 0.2, (n, 3)))
    qlo, qhi = qc - qe, qc + qe
    tc = f32(qc + rng.normal(scale=0.4, size=(n, 3))); te = f32(rng.uniform(0, 0.2, (n, 3)))
    tlo, thi = tc - te, tc + te
    g = np.maximum(np.float32(0), np.maximum(tlo - qhi, qlo - thi))
    bg2 = (g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]      # box_gap2: plain mul/add association
    for _ in range(8):
        p = np.clip((qlo + f32(rng.random((n, 3))) * (qhi - qlo)).astype(np.float32), qlo, qhi)
        q = np.clip((tlo + f32(rng.random((n, 3))) * (thi - tlo)).astype(np.float32), tlo, thi)
        d2 = canon_d2(p, q)
        # different association (three rounded products and two rounded adds vs mul + two fmas): a few ulps apart, which is
        # what the 1.00001 factor on the threshold covers -- the bound holds with that margin
        assert np.all(bg2 <= d2 * np.float32(1.00001) + np.float32(1e-30))


# ------------------------------------------------------------------ the projective window search (S3D_PROJ_SEARCH)
def _proj_c(fx, fy, cx, cy, W, H):
    """the constant of slam3d_icp_create (icp_capi.hip): fmax * sqrt(1 + amax^2 + bmax^2) * 1.001, as a float"""
    am = max(abs((0.0 - cx) / fx), abs((W - 1.0 - cx) / fx)); bm = max(abs((0.0 - cy) / fy), abs((H - 1.0 - cy) / fy))
    return np.float32(max(fx, fy) * np.sqrt(1.0 + am * am + bm * bm) * 1.001)


@pytest.mark.parametrize("seed,W,H,fx,fy,cx,cy", [(0, 160, 120, 131.25, 131.25, 79.875, 58.875), (1, 96, 72, 90.0, 70.0, 40.0, 30.0),
                                                  (2, 128, 96, 60.0, 60.0, 100.0, 10.0), (3, 160, 120, 525.0, 525.0, 79.5, 59.5)])
def test_projective_window_covers_every_closer_target(seed, W, H, fx, fy, cx, cy):
    """icp_kernels.hpp, projective window search: for a query p' (z' > 0.05) with bound U, every target of a back-projected
    depth image that lies OUTSIDE the (2r+1)^2 window around the rounded projection of p', r = ceil(proj_c * (sqrt(U) +
    1e-5) / z' - 0.49), is strictly farther than sqrt(U) -- so the window alone holds the exact neighbour and every tie.
    Checked against the oracle's own back-projection (float coordinates as the GPU stores them), with the hardware's
    approximate reciprocal / square root emulated at their worst (1 ulp towards the smaller window), including
    off-centre principal points, fx != fy, grazing depths and queries that project outside the image."""
    import oracle_lib as O
    from types import SimpleNamespace
    rng = np.random.default_rng(100 + seed)
    intr = SimpleNamespace(width=W, height=H, fx=fx, fy=fy, cx=cx, cy=cy, depth_factor=1000.0)
    p = O.params(intr)
    depth = rng.integers(300, 7000, (H, W)).astype(np.uint16)          # every pixel valid, depths uncorrelated: the worst case
    depth[rng.random((H, W)) < 0.05] = 0
    cloud = O.backproject(depth, p).reshape(-1, 4)
    tv = np.isfinite(cloud[:, 0])
    q = cloud[:, :3].astype(np.float64)
    uu, vv = np.meshgrid(np.arange(W), np.arange(H)); uu = uu.ravel(); vv = vv.ravel()
    c = _proj_c(fx, fy, cx, cy, W, H)
    n = 600
    # queries: near targets (small bounds), random points of the frustum and beyond it, bounds from sub-millimetre to the gate
    base = q[rng.choice(np.nonzero(tv)[0], n)]
    pq = f32(base + rng.normal(0, 1, (n, 3)) * rng.choice([1e-4, 1e-3, 1e-2, 0.1], (n, 1)))
    pq[: n // 6] = f32(np.c_[rng.uniform(-6, 6, n // 6), rng.uniform(-5, 5, n // 6), rng.uniform(0.06, 8, n // 6)])
    U = f32(rng.choice([1e-8, 1e-6, 2.5e-5, 1e-4, 1e-3, 1e-2], n) * rng.uniform(0.3, 1.0, n))
    ok = pq[:, 2] > np.float32(0.05)
    one_ulp = np.float32(1.0 - 2.0 ** -22)
    izp = (np.float32(1.0) / pq[:, 2]) * one_ulp                       # v_rcp_f32, 1 ulp low
    uf = f32(np.float32(fx) * pq[:, 0] * izp + np.float32(cx)); vf = f32(np.float32(fy) * pq[:, 1] * izp + np.float32(cy))
    rn = c * izp * (np.sqrt(U) * one_ulp + np.float32(1e-5)) - np.float32(0.49)
    r = np.where(rn > 0, np.ceil(rn), 0).astype(np.int64)
    checked = 0
    for k in np.nonzero(ok & (r <= 6) & (np.abs(uf) < 1e6) & (np.abs(vf) < 1e6))[0]:
        u0, v0 = int(np.rint(uf[k])), int(np.rint(vf[k]))
        outside = tv & ((np.abs(uu - u0) > r[k]) | (np.abs(vv - v0) > r[k]))
        d2 = ((q[outside] - pq[k].astype(np.float64)) ** 2).sum(1)
        assert d2.size == 0 or d2.min() > float(U[k]) * (1.0 + 1e-6), (k, r[k], float(U[k]), float(d2.min()))
        checked += 1
    assert checked > 100

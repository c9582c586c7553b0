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

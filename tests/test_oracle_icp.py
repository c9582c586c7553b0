"""Oracle self-consistency and known-answer tests (CPU only)."""
import numpy as np
import pytest

import oracle_lib as O
from slam3d_gx_amd import synth


def _clouds(pr):
    return synth.backproject_numpy(pr.depth_src, pr.intr), synth.backproject_numpy(pr.depth_tgt, pr.intr)


def test_backproject_oracle_equals_numpy_spec():
    pr = synth.make_pair(1000, 320, 240)
    a = O.backproject(pr.depth_src, O.params(pr.intr))
    b = synth.backproject_numpy(pr.depth_src, pr.intr)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))


def test_backproject_zfilter_and_zero_are_invalid():
    intr = synth.Intrinsics.scaled(8, 4)
    d = np.zeros((4, 8), np.uint16); d[0, 0] = 7000; d[0, 1] = 7001; d[0, 2] = 1
    c = O.backproject(d, O.params(intr))
    assert c[0, 0, 2] == np.float32(7.0) and np.isnan(c[0, 1, 2]) and c[0, 2, 2] == np.float32(0.001)
    assert np.isnan(c[1:, :, :3]).all() and (c[1:, :, 3] == 0).all()


@pytest.mark.parametrize("estimator", [0, 1])
@pytest.mark.parametrize("seed", [1000, 1001])
def test_kdtree_equals_bruteforce(estimator, seed):
    pr = synth.make_pair(seed, 160, 120)
    s4, t4 = _clouds(pr)
    rb = O.icp(s4, t4, O.params(pr.intr, estimator=estimator, iterations=4, nn_method=0))
    rk = O.icp(s4, t4, O.params(pr.intr, estimator=estimator, iterations=4, nn_method=1))
    assert np.array_equal(rb["idx"], rk["idx"]) and np.array_equal(rb["d2"], rk["d2"])
    assert np.array_equal(rb["T_trace"], rk["T_trace"]) and np.array_equal(rb["sums_trace"], rk["sums_trace"])


def test_thread_count_does_not_change_results():
    pr = synth.make_pair(1002, 160, 120)
    s4, t4 = _clouds(pr)
    r1 = O.icp(s4, t4, O.params(pr.intr, iterations=3, threads=1))
    r4 = O.icp(s4, t4, O.params(pr.intr, iterations=3, threads=4))
    assert np.array_equal(r1["T_trace"], r4["T_trace"]) and np.array_equal(r1["idx"], r4["idx"])


def test_tie_break_lowest_index_and_gate():
    """Duplicate target points: the smaller linear index must win; beyond max_corr_dist => -1."""
    intr = synth.Intrinsics.scaled(8, 2)
    src = np.full((2, 8, 4), np.nan, np.float32); tgt = np.full((2, 8, 4), np.nan, np.float32)
    src[0, 0, :3] = [0.0, 0.0, 1.0]
    src[0, 1, :3] = [2.0, 0.0, 1.0]           # far from everything (> 0.1 m) -> -1
    tgt[0, 3, :3] = [0.01, 0.0, 1.0]
    tgt[0, 5, :3] = [0.01, 0.0, 1.0]          # exact duplicate, higher index
    tgt[1, 2, :3] = [-0.01, 0.0, 1.0]         # same distance on the other side, even higher index
    for method in (0, 1):
        idx, d2, ns = O.nn_once(src, tgt, O.params(intr, nn_method=method))
        assert ns == 2 and idx[0] == 3 and idx[1] == -1 and np.isinf(d2[1])
        assert d2[0] == np.float32(0.01) * np.float32(0.01)
        assert (idx[2:] == -1).all()


def test_known_answer_noise_free():
    """Noise-free scene without holes: point-to-plane ICP recovers the analytic pose to the
    quantisation floor of the u16-millimetre depth (documented bound)."""
    pr = synth.make_pair(1000, 320, 240, noise=False, holes=False)
    s4, t4 = _clouds(pr)
    r = O.icp(s4, t4, O.params(pr.intr, iterations=30))
    rot, tr = O.pose_error(pr.T_gt, r["T"])
    assert r["status"] == 0
    assert rot < 5e-4 and tr < 2e-3


def test_identity_pair_stays_at_identity():
    pr = synth.make_pair(1000, 160, 120)
    s4, _ = _clouds(pr)
    r = O.icp(s4, s4, O.params(pr.intr, iterations=3, estimator=1))
    # every valid point matches itself at distance zero
    valid = np.isfinite(s4[..., 2]).reshape(-1)
    assert np.array_equal(r["idx"][valid], np.nonzero(valid)[0]) and (r["idx"][~valid] == -1).all()
    assert (r["d2"][valid] == 0).all() and r["rmse"] == 0.0
    rot, tr = O.pose_error(np.eye(4), r["T_trace"][-1])
    assert rot < 1e-7 and tr < 1e-12


def test_failure_modes():
    pr = synth.make_pair(1000, 160, 120)
    s4, t4 = _clouds(pr)
    empty = np.full_like(t4, np.nan)
    r = O.icp(s4, empty, O.params(pr.intr, iterations=2))
    assert r["status"] == 1 and np.array_equal(r["T"], np.eye(4)) and r["inliers"] == 0
    r = O.icp(empty, t4, O.params(pr.intr, iterations=2))
    assert r["status"] == 1 and r["n_src"] == 0
    # norm threshold: src/GraphicEnd.cpp:621
    r = O.icp(s4, t4, O.params(pr.intr, iterations=5, error_threshold=1e-6))
    assert r["status"] == 2 and np.array_equal(r["T"], np.eye(4)) and r["norm"] > 1e-6
    # a single plane constrains 3 of 6 DoF -> damped solve -> DEGENERATE
    intr = synth.Intrinsics.scaled(64, 48)
    d = np.full((48, 64), 2000, np.uint16)
    c = synth.backproject_numpy(d, intr)
    r = O.icp(c, c, O.params(intr, iterations=2))
    assert r["status"] == 3 and np.array_equal(r["T"], np.eye(4))


def test_fit_planes_synthetic_room():
    pr = synth.make_pair(1000, 320, 240)
    hb = max(2, int(round(32 * 320 / 640.0)))
    d, lab = synth.render_depth(np.eye(4), pr.intr, 1000, 1, hole_block=hb, want_labels=True)
    assert np.array_equal(d, pr.depth_src)
    c = synth.backproject_numpy(d, pr.intr)
    planes, counts = O.fit_planes(c, lab, 3)
    expect = np.array([[0, -1, 0, 1.2], [1, 0, 0, 2.0], [0, 0, -1, 4.5]])   # d >= 0 (src/GraphicEnd.cpp:383-387)
    assert (counts > 100).all()
    assert np.allclose(planes, expect, atol=5e-3)
    assert (planes[:, 3] >= 0).all()


def test_sums_are_fixed_point_integers_and_order_free():
    """Spec S4 summation (round 4: integer Gram totals): every derived sum is an integer times its power of two -- A^T A
    entries multiples of 2^-40 (the finest pair of scales: n n at 2^-20 each), A^T b of 2^-(20 + EB), sum b^2 of 2^-2EB, the
    svd estimator's of 2^-32 --, the count is an integer, and a re-threaded evaluation gives the same bits (integer
    addition is associative)."""
    pr = synth.make_pair(1003, 160, 120)
    s4 = synth.backproject_numpy(pr.depth_src, pr.intr)
    t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
    eb = 23                                                  # max_corr_dist 0.10 = 0.8 * 2^-3 -> 20 - (-3)
    for est in (0, 1):
        r1 = O.icp(s4, t4, O.params(pr.intr, estimator=est, iterations=3, threads=1))
        r5 = O.icp(s4, t4, O.params(pr.intr, estimator=est, iterations=3, threads=5))
        assert np.array_equal(r1["sums_trace"], r5["sums_trace"]) and np.array_equal(r1["T_trace"], r5["T_trace"])
        S = r1["sums_trace"]
        unit = np.full(29, 2.0 ** 32)
        if est == 0:
            unit[:21] = 2.0 ** 40; unit[21:27] = 2.0 ** (20 + eb); unit[28] = 2.0 ** (2 * eb)
        unit[27] = 1.0
        q = S * unit
        assert np.array_equal(q, np.rint(q)) and np.abs(q).max() < 2.0 ** 62
        assert r1["inliers"] == int(S[-1, 27])


BASELINE_MD = dict(noise_sigma=0.0012, hole_block=8, hole_prob=0.25)


def _kinect_depth(name):
    import os
    from PIL import Image
    return np.array(Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kinect", name))).astype(np.uint16)


@pytest.mark.parametrize("workload", ["low_noise", "baseline_md"])
def test_coarse_iterations_do_not_move_the_final_pose(workload):
    """Spec S4c deviates from SURVEY.md App. C3 ("for each valid source i"): iterations 0-2 take a quarter of the sources.  The
    deviation is BOUNDED here: after 20 iterations the pose of coarse_iterations = 3 and the pose of
    coarse_iterations = 0 differ by at most 1e-4 rad / 1e-4 m -- the metric's own bar -- on seeds 1000..1003 under both synthetic
    workloads (measured: <= 3.5e-6 rad / 4.7e-6 m)."""
    kw = BASELINE_MD if workload == "baseline_md" else {}
    worst = (0.0, 0.0)
    for seed in (1000, 1001, 1002, 1003):
        pr = synth.make_pair(seed, **kw)
        s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
        T = {c: O.icp(s4, t4, O.params(pr.intr, iterations=20, nn_method=1, coarse_iterations=c))["T_trace"][-1] for c in (0, 3)}
        rot, tr = O.pose_error(T[0], T[3])
        assert rot <= 1e-4 and tr <= 1e-4, (seed, rot, tr)
        worst = (max(worst[0], rot), max(worst[1], tr))
    assert worst[0] < 2e-5 and worst[1] < 2e-5, worst


def test_coarse_iterations_on_the_reference_kinect_frames():
    """The same bound on real frames: the perturbed self-alignments (which converge) end within 1e-4 rad / 1e-4 m whatever
    coarse_iterations is; the wide-baseline pair dep1 -> dep2 does NOT converge in 20 iterations (no unique
    answer from the identity) -- its two poses differ by less than the distance the pose still moves in its last iteration, which
    is all that can be asked of a run that has no fixed point yet.  Both estimators."""
    d1, d2 = _kinect_depth("exp1_dep_1.png"), _kinect_depth("exp1_dep_2.png")
    intr = synth.Intrinsics()
    Ti = synth.pose_from_seed(77, 2.0, 0.03)
    for est in (0, 2):
        for a, b, T0 in ((d1, d1, Ti), (d2, d2, Ti), (d1, d2, None)):
            r = {}
            for c in (0, 3):
                p = O.params(intr, iterations=20, nn_method=1, coarse_iterations=c, estimator=est, plane_pair_gate=1 if est == 2 else 0)
                r[c] = O.icp(O.backproject(a, p), O.backproject(b, p), p, T_init=T0)["T_trace"]
            rot, tr = O.pose_error(r[0][-1], r[3][-1])
            srot, str_ = O.pose_error(r[3][-2], r[3][-1])
            if T0 is not None and est == 0:
                assert rot <= 1e-4 and tr <= 1e-4, (est, rot, tr)
            else:
                # no fixed point yet: the wide-baseline pair, and (round 6, segmentation threshold 0.04 m) the plane estimator's
                # dep2 -> dep2, which still moves 0.1 mm per iteration at its end -- the two runs may differ by what a run still moves
                k = 4.0 if est == 2 else 1.0      # (the plane estimator's wide-baseline run moves 2 mm per iteration at its end and ends NORM_EXCEEDED either way)
                assert rot <= max(1e-4, 1.5 * k * srot) and tr <= max(1e-4, 3.0 * k * str_), (est, rot, tr, srot, str_)

"""SLAM3D_EST_PLANE -- plane-ICP proper (DESIGN.md spec S2p / S4p; SURVEY.md App. C2 "points take their plane's normal";
src/GraphicEnd.cpp:353-430 planes per frame, :459-484 association, :557-659 pose from plane-wise correspondences).

CPU part: the oracle's per-plane normals and association against plain numpy; properties of the estimator.
GPU part (-m gpu): the HIP path against the oracle, bit for bit, through the C-ABI.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from slam3d_gx_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
BASELINE_MD = dict(noise_sigma=0.0012, hole_block=8, hole_prob=0.25)      # BASELINE.md section 4's synthetic workload


def _pair(seed, w=640, h=480, **kw):
    pr = synth.make_pair(seed, w, h, **kw)
    return pr, synth.backproject_numpy(pr.depth_src, pr.intr), synth.backproject_numpy(pr.depth_tgt, pr.intr)


def _kinect(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(HERE, "golden", "kinect", name))).astype(np.uint16)


# ------------------------------------------------------------------------------------------------ CPU: the oracle's S2p
@pytest.mark.parametrize("plane_only", [0, 1])
def test_plane_normals_are_the_segmentation_plus_the_window_normals(plane_only):
    """orc_plane_normals == (orc_segment_planes labels -> the plane's (a, b, c), w = 1 + plane) over (orc_normals, w = 0.75)"""
    pr, s4, t4 = _pair(1000, 320, 240)
    p = O.params(pr.intr, estimator=2, plane_only=plane_only)
    nrm, planes, labels = O.plane_normals(t4, p)
    pl, lab = O.segment_planes(t4, zmax=7.0, distance_threshold=p.seg_distance_threshold, plane_percent=p.seg_plane_percent,
                               max_planes=p.seg_max_planes, hypotheses=p.seg_hypotheses, seed=p.seg_seed)
    assert np.array_equal(lab, labels) and len(pl) == len(planes) >= 2
    win = O.normals(t4, p).reshape(-1, 4)
    want = np.zeros((lab.size, 4), dtype=np.float32)
    if not plane_only:
        m = win[:, 3] > 0.5
        want[m, :3] = win[m, :3]; want[m, 3] = 0.75
    for r, q in enumerate(pl):
        m = lab == r
        want[m, :3] = q["coeff"][:3]; want[m, 3] = 1 + r
        assert q["coeff"][3] >= 0 and abs(np.linalg.norm(q["coeff"][:3].astype(np.float64)) - 1) < 1e-6
        # d >= 0  <=>  the normal looks toward the camera for the plane's points
        pts = t4.reshape(-1, 4)[m, :3].astype(np.float64)
        assert (pts @ q["coeff"][:3].astype(np.float64) < 0.09).all()
    assert np.array_equal(nrm.reshape(-1, 4), want)


def test_plane_association_is_the_exact_nearest_plane_under_the_initial_pose():
    rng = np.random.default_rng(5)
    for trial in range(50):
        n1, n2 = rng.integers(0, 5), rng.integers(0, 5)
        def planes(n):
            P = np.zeros((n, 8), dtype=np.float32)
            nv = rng.normal(size=(n, 3)); nv /= np.linalg.norm(nv, axis=1, keepdims=True)
            P[:, :3] = nv; P[:, 3] = rng.uniform(0.5, 5.0, n)
            return P
        A, B = planes(n1), planes(n2)
        T = synth.pose_from_seed(100 + trial, 10.0, 0.3) if trial % 2 else None
        got = O.plane_assoc(A, B, T)
        Tm = np.eye(4) if T is None else T
        for i in range(n1):
            n = Tm[:3, :3] @ A[i, :3].astype(np.float64)
            d = float(A[i, 3]) - n @ Tm[:3, 3]
            if d < 0:
                n, d = -n, -d
            m = np.array([*n, d], dtype=np.float32)
            if n2 == 0:
                assert got[i] == -1
                continue
            d2 = ((B[:, :4].astype(np.float64) - m.astype(np.float64)) ** 2).sum(1)
            assert got[i] == int(np.argmin(d2)) or np.sort(d2)[1] - np.sort(d2)[0] < 1e-6
        # and it is what the library's host-side gate computes (slam3d_plane_gate shares the arithmetic)


def test_plane_estimator_settles_where_the_window_estimator_keeps_moving():
    """The measured reason for the estimator: on the reference's Kinect frames the window-normal run still
    moves by a millimetre per iteration at its end, the plane run has stopped -- so its late launches track and certify.
    Round 6: the test also looks at WHERE the run stops.  The perturbed self-alignments (2 degrees / 3 cm off)
    of BOTH frames must come back to the identity: dep1 within 1 mrad / 2 mm; dep2 within 3 mrad / 5 mm -- point-to-plane residuals
    do not see a slide along the frame's planes and the identity is one fixed point among several, its rotation error stays at
    2.0-2.6 mrad for every segmentation threshold between 0.02 and 0.08 m (sweep in DESIGN.md section 3).  The default threshold
    (0.04 m, chosen from that sweep) is what is under test: at the reference's plane-extraction key (0.08 m) dep2 ended 10.7 mm off."""
    intr = synth.Intrinsics()
    Ti = synth.pose_from_seed(77, 2.0, 0.03)
    bars = {"exp1_dep_1.png": (1e-3, 2e-3), "exp1_dep_2.png": (3e-3, 5e-3)}
    for name, (rot_bar, tr_bar) in bars.items():
        d = _kinect(name)
        step, end = {}, {}
        for est in (0, 2):
            p = O.params(intr, iterations=20, estimator=est, plane_pair_gate=1 if est == 2 else 0)
            assert est == 0 or abs(p.seg_distance_threshold - 0.04) < 1e-7
            c = O.backproject(d, p)
            r = O.icp(c, c, p, T_init=Ti)
            assert r["status"] == 0
            step[est] = O.pose_error(r["T_trace"][15], r["T_trace"][16])
            end[est] = O.pose_error(np.eye(4), r["T_trace"][-1])
        assert step[2][1] < 2e-4 and step[2][0] < 1e-4, (name, step)        # (0.1 mm per iteration on dep2, 40 times less than the window run)
        assert end[2][0] <= rot_bar and end[2][1] <= tr_bar, (name, end)
        assert end[2][1] < end[0][1], (name, end)          # closer to the identity than the window estimator's run (4.1 mm on both frames)
        if name == "exp1_dep_2.png":
            assert step[0][1] > 1e-3                        # window normals: > 1 mm per iteration at iteration 15
            p8 = O.params(intr, iterations=20, estimator=2, plane_pair_gate=1, seg_distance_threshold=0.08)
            c = O.backproject(d, p8)
            far = O.pose_error(np.eye(4), O.icp(c, c, p8, T_init=Ti)["T_trace"][-1])
            assert far[1] > 2.0 * end[2][1], (far, end)     # the old default stops twice as far away


def test_planes_only_is_degenerate_on_the_synthetic_room_and_says_so():
    """Why pixels on no plane keep their window normal by default: the synthetic room's target frame yields three z-facing
    planes, the 6x6 system needs damping, and the run must never report OK."""
    pr, s4, t4 = _pair(1000)
    r = O.icp(s4, t4, O.params(pr.intr, iterations=6, estimator=2, plane_only=1))
    assert r["status"] in (1, 3) and np.array_equal(r["T"], np.eye(4))      # DEGENERATE at the 0.08 m threshold of round 5, TOO_FEW_INLIERS at 0.04: never OK
    r8 = O.icp(s4, t4, O.params(pr.intr, iterations=6, estimator=2, plane_only=1, seg_distance_threshold=0.08))
    assert r8["status"] == 3 and np.array_equal(r8["T"], np.eye(4))
    r = O.icp(s4, t4, O.params(pr.intr, iterations=20, estimator=2))
    rot, tr = O.pose_error(pr.T_gt, r["T"])
    assert r["status"] == 0 and rot < 2e-3 and tr < 5e-3


# ------------------------------------------------------------------------------------------------ GPU: HIP == oracle
def _gpu_case(h, pr, s4, t4, ro, depth, iters, T_init=None):
    kw = {} if T_init is None else {"T_init": [T_init]}
    rg = h.align_depth_batch([pr.depth_src], [pr.depth_tgt], **kw)[0] if depth else h.align(s4, t4, T_init)
    idx, d2 = h.get_correspondences(0)
    Tt, St = h.get_trace(0)
    assert rg["n_src"] == ro["n_src"] and rg["n_tgt"] == ro["n_tgt"], (rg["n_tgt"], ro["n_tgt"])
    assert np.array_equal(idx, ro["idx"]), f"depth={depth}: {(idx != ro['idx']).sum()} index mismatches"
    assert np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32))
    assert np.array_equal(St[:iters], ro["sums_trace"]) and np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"])
    assert rg["inliers"] == ro["inliers"] and rg["status"] == ro["status"]
    return rg


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
@pytest.mark.parametrize("size", [(160, 120), (320, 240)])
def test_small_plane_icp_vs_bruteforce_oracle(gpu_lib, size, flags):
    from slam3d_gx_amd import capi
    pr, s4, t4 = _pair(1000, *size)
    ro = O.icp(s4, t4, O.params(pr.intr, estimator=2, iterations=6, nn_method=0, plane_pair_gate=flags & 1, plane_only=(flags >> 1) & 1))
    with capi.IcpHandle(capi.default_params(pr.intr, estimator=capi.EST_PLANE, iterations=6, plane_flags=flags)) as h:
        _gpu_case(h, pr, s4, t4, ro, False, 6)
        _, _, nrm = h.get_clouds(0, normals=True)
        want, planes, _ = O.plane_normals(t4, O.params(pr.intr, estimator=2, plane_only=(flags >> 1) & 1))
        assert np.array_equal(nrm, want)
        got = h.get_frame_planes(1)
        assert len(got) == len(planes) and all(np.array_equal(g["coeff"], q[:4]) and g["count"] == int(q[7]) for g, q in zip(got, planes))
        if flags & 1:
            _, splanes, _ = O.plane_normals(s4, O.params(pr.intr, estimator=2))
            assert np.array_equal(h.get_plane_assoc(0)[: len(splanes)], O.plane_assoc(splanes, planes))
        _gpu_case(h, pr, s4, t4, ro, True, 6)           # the same pair as depth images: window search, same bits


@pytest.mark.gpu
@pytest.mark.parametrize("seed,gate", [(1000, 1), (1001, 1), (1000, 0)])
def test_full_640x480_plane_icp_baseline_md_workload(gpu_lib, seed, gate):
    """640x480 x 20 under BASELINE.md section 4's workload, depth and cloud inputs, graph and direct launches:
    every iterate, every sum, the last indices and d2 bit-identical to the kd-tree oracle; index parity per iteration."""
    from slam3d_gx_amd import capi
    pr, s4, t4 = _pair(seed, **BASELINE_MD)
    po = O.params(pr.intr, estimator=2, iterations=20, nn_method=1, plane_pair_gate=gate)
    ro = O.icp(s4, t4, po)
    assert ro["status"] == 0 and ro["n_tgt"] > 0.6 * ro["n_src"]          # the planes give most targets a normal (window normals: a fifth)
    with capi.IcpHandle(capi.default_params(pr.intr, estimator=capi.EST_PLANE, iterations=20, plane_flags=gate)) as h:
        for depth, trace in ((True, False), (False, False), (True, True)):
            h.set_corr_trace(trace)
            rg = _gpu_case(h, pr, s4, t4, ro, depth, 20)
            if trace and not gate:
                for it in (0, 2, 3, 10, 19):
                    want, _, _ = O.nn_once(s4, t4, po, T=ro["T_trace"][it], use_normals=2, coarse=(it < 3))
                    assert np.array_equal(h.get_correspondences_at(it), want), it
    rot_gt, tr_gt = O.pose_error(pr.T_gt, rg["T"])
    assert rot_gt < 1e-2 and tr_gt < 3e-2, (rot_gt, tr_gt)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["dep1->dep2", "dep1->dep1", "dep2->dep2"])
def test_plane_icp_on_the_reference_kinect_frames(gpu_lib, case):
    from slam3d_gx_amd import capi
    d = {"1": _kinect("exp1_dep_1.png"), "2": _kinect("exp1_dep_2.png")}
    a, b = case[3], case[9]
    intr = synth.Intrinsics()
    pr = synth.FramePair(-1, intr, d[a], d[b], np.eye(4))
    s4, t4 = synth.backproject_numpy(d[a], intr), synth.backproject_numpy(d[b], intr)
    Ti = synth.pose_from_seed(77, 2.0, 0.03) if a == b else None
    ro = O.icp(s4, t4, O.params(intr, estimator=2, iterations=20, nn_method=1, plane_pair_gate=1), T_init=Ti)
    with capi.IcpHandle(capi.default_params(intr, estimator=capi.EST_PLANE, iterations=20, plane_flags=capi.PLANE_PAIR_GATE)) as h:
        for depth in (True, False):
            _gpu_case(h, pr, s4, t4, ro, depth, 20, Ti)
    if a == b:      # the perturbed self-alignment has stopped moving long before the run ends
        assert max(O.pose_error(ro["T_trace"][15], ro["T_trace"][16])) < 2e-4


@pytest.mark.gpu
def test_plane_icp_batch_frames_seg_params_and_every_nn_mode(gpu_lib):
    """A batch (the batched segmentation path: five launches per round), resident frames shared between pairs, other
    segmentation parameters, and the full-scan kernels: all bit-identical to the oracle / to the single-pair runs."""
    from slam3d_gx_amd import capi
    seeds = [1000, 1001, 1002, 1003]
    prs = [_pair(s, 320, 240) for s in seeds]
    intr = prs[0][0].intr
    ros = [O.icp(s4, t4, O.params(intr, estimator=2, iterations=6, nn_method=1, plane_pair_gate=1)) for _, s4, t4 in prs]
    with capi.IcpHandle(capi.default_params(intr, estimator=capi.EST_PLANE, iterations=6, max_batch=4, plane_flags=1)) as h:
        rs = h.align_batch([p[1] for p in prs], [p[2] for p in prs])
        for b, (ro, rg) in enumerate(zip(ros, rs)):
            Tt, St = h.get_trace(b)
            assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:6], ro["sums_trace"]), b
            assert np.array_equal(h.get_correspondences(b)[0], ro["idx"]) and rg["status"] == ro["status"]
        # other segmentation parameters: the frames are rebuilt
        sp = h.seg_params(distance_threshold=0.03, hypotheses=32, seed=9)
        h.set_seg_params(sp)
        ro2 = O.icp(prs[0][1], prs[0][2], O.params(intr, estimator=2, iterations=6, nn_method=1, plane_pair_gate=1, seg_distance_threshold=0.03,
                                                   seg_hypotheses=32, seg_seed=9))
        h.set_clouds_host(0, prs[0][1], prs[0][2])
        h.run(1); h.fetch_results(1)
        assert np.array_equal(h.get_trace(0)[0].reshape(-1, 4, 4), ro2["T_trace"]) and not np.array_equal(ro2["T_trace"], ros[0]["T_trace"])
    for mode in (capi.NN_BRUTE_VALU, capi.NN_BRUTE_MFMA):
        pr, s4, t4 = prs[1]
        with capi.IcpHandle(capi.default_params(intr, estimator=capi.EST_PLANE, iterations=6, plane_flags=1, nn_mode=mode)) as h:
            h.align(s4, t4)
            assert np.array_equal(h.get_trace(0)[0].reshape(-1, 4, 4), ros[1]["T_trace"]), mode
            assert np.array_equal(h.get_correspondences(0)[0], ros[1]["idx"])


@pytest.mark.gpu
@pytest.mark.parametrize("min_cos", [0.0, 0.5])
def test_plane_icp_resident_frames_change_roles(gpu_lib, min_cos):
    """Resident frames under the pair gate: a frame that is only a SOURCE is segmented but gets no window normals (labels are all the
    gate needs -- unless min_normal_cos compares source normals); the same frame as a TARGET later must then be completed, a former
    target serves as a source as it is, two pairs of one launch share frames in both roles, and a frame whose contents are replaced
    is rebuilt.  Every run equals the oracle's run on the two clouds, whatever the frames were used for before."""
    from slam3d_gx_amd import capi
    (pr, a, b), (_, c, d) = _pair(1000, 320, 240), _pair(1001, 320, 240)
    clouds = {0: a, 1: b, 2: c}
    po = O.params(pr.intr, estimator=2, iterations=5, nn_method=1, plane_pair_gate=1, min_normal_cos=min_cos)
    want = {}
    def oracle(i, j):
        if (i, j) not in want:
            want[(i, j)] = O.icp(clouds[i], clouds[j], po)
        return want[(i, j)]
    with capi.IcpHandle(capi.default_params(pr.intr, estimator=capi.EST_PLANE, iterations=5, max_batch=2, extra_frames=3, plane_flags=capi.PLANE_PAIR_GATE,
                                            min_normal_cos=min_cos)) as h:
        f0 = h.first_free_frame()
        for k, cl in clouds.items():
            h.frame_set_cloud_host(f0 + k, cl)
        def run(pairs):
            for slot, (i, j) in enumerate(pairs):
                h.set_pair(slot, f0 + i, f0 + j)
            h.run(len(pairs)); rs = h.fetch_results(len(pairs))
            for slot, (i, j) in enumerate(pairs):
                ro = oracle(i, j)
                Tt, St = h.get_trace(slot)
                assert np.array_equal(Tt, ro["T_trace"]) and np.array_equal(St, ro["sums_trace"]), (pairs, slot)
                assert rs[slot]["status"] == ro["status"] and rs[slot]["inliers"] == ro["inliers"]
        run([(0, 1)])                   # 0: source only, 1: target
        run([(1, 2)])                   # the former target as a source
        run([(2, 0)])                   # the labels-only source as a target: its normals are completed now
        run([(0, 1), (1, 0)])           # both roles in one launch
        _, planes, _ = O.plane_normals(clouds[0], O.params(pr.intr, estimator=2))
        got = h.get_frame_planes(f0)
        assert len(got) == len(planes) and all(np.array_equal(g["coeff"], q[:4]) for g, q in zip(got, planes))
        clouds[0] = d                   # new contents for frame 0: everything of it is rebuilt
        want.clear()
        h.frame_set_cloud_host(f0, d)
        run([(0, 2)])
        run([(1, 0)])


@pytest.mark.gpu
@pytest.mark.parametrize("gate", [0, 1])
def test_plane_icp_in_dense_mode_over_row_shards(gpu_lib, gate):
    """SLAM3D_EST_PLANE in the dense mode (one pair, the SOURCE rows sharded over ranks; both frames are segmented whole on every
    rank: the planes are global, only the rows a rank accumulates are its own): two emulated ranks exchanging integer totals on the
    host, and the single-call run over a host all-reduce, equal the unsharded oracle bit for bit."""
    from slam3d_gx_amd import capi, dense, shard
    pr, s4, t4 = _pair(1002, 320, 240)
    iters = 6
    ro = O.icp(s4, t4, O.params(pr.intr, estimator=2, iterations=iters, nn_method=1, plane_pair_gate=gate))
    kw = dict(estimator=capi.EST_PLANE, plane_flags=gate, iterations=iters)
    hs = [capi.IcpHandle(capi.default_params(pr.intr, **kw)) for _ in range(2)]
    try:
        for r, h in enumerate(hs):
            h.set_clouds_host(0, s4, t4)
            h.dense_set_rows(*shard.dense_row_range(pr.intr.height, 2, r))
            h.dense_begin(None)
        total = None
        for _ in range(iters):
            parts = [h.dense_partial() for h in hs]
            total = parts[0] + parts[1]
            for h in hs:
                h.dense_update(total)
        res = [h.dense_finish(total) for h in hs]
    finally:
        for h in hs:
            h.close()
    assert np.array_equal(res[0]["T_raw"], res[1]["T_raw"]) and np.array_equal(res[0]["T_raw"], ro["T_trace"][-1])
    assert res[0]["inliers"] == ro["inliers"] and res[0]["status"] == ro["status"]
    with capi.IcpHandle(capi.default_params(pr.intr, **kw)) as h:
        h.set_clouds_host(0, s4, t4)
        calls = []
        def allreduce(d_buf, count, stream):          # one rank: the sum over the ranks is the buffer itself
            calls.append(count)
            return 0
        r1 = h.dense_run_with(0, 1, allreduce)
    assert np.array_equal(r1["T_raw"], ro["T_trace"][-1]) and r1["inliers"] == ro["inliers"]


@pytest.mark.gpu
def test_handles_give_back_every_device_byte(gpu_lib):
    """Create / use / destroy, many times, with everything round 5 added in use (plane estimator + gate, batched voxel grid,
    segmentation, full-scan modes, unorganized handles, stamping): the device's free memory returns to where it was -- no buffer
    of a handle outlives slam3d_icp_destroy."""
    import torch
    from slam3d_gx_amd import capi
    pr, s4, t4 = _pair(1000, 160, 120)
    pts = np.ascontiguousarray(s4.reshape(-1, 4))
    lst = np.full((1, 3000, 4), np.nan, np.float32); lst[0, :2500, :3] = pts[np.isfinite(pts[:, 2])][:2500, :3]; lst[0, :2500, 3] = 1.0

    def cycle():
        with capi.IcpHandle(capi.default_params(pr.intr, estimator=capi.EST_PLANE, plane_flags=capi.PLANE_PAIR_GATE, iterations=4, max_batch=2,
                                                extra_frames=2)) as h:
            h.set_stamping(8)
            h.align_batch([s4, t4], [t4, s4])
            h.get_frame_planes(0); h.get_plane_assoc(0)
            h.segment_planes(s4)
            h.voxel_grid(pts)
            d = torch.from_numpy(pts).to("cuda:0"); o = [torch.empty_like(d) for _ in range(3)]
            h.voxel_grid_batch_device([d.data_ptr()] * 3, [pts.shape[0]] * 3, [x.data_ptr() for x in o])
            del d, o
        for mode in (capi.NN_BRUTE_MFMA, capi.NN_BRUTE_VALU):
            with capi.IcpHandle(capi.default_params(pr.intr, iterations=2, nn_mode=mode)) as h:
                h.align(s4, t4)
        with capi.IcpHandle(capi.default_params(synth.Intrinsics(width=3000, height=1), estimator=capi.EST_SVD, iterations=2)) as h:
            h.align(lst, lst)
        torch.cuda.synchronize(); torch.cuda.empty_cache()

    cycle(); cycle()                                   # (first uses: module load, constant tables, the runtime's own pools)
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(12):
        cycle()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), (free0, free1)   # twelve cycles of ~100 MB of handles each: nothing accumulates


@pytest.mark.gpu
def test_plane_flags_are_validated(gpu_lib):
    from slam3d_gx_amd import capi
    intr = synth.Intrinsics.scaled(160, 120)
    for bad in (dict(estimator=capi.EST_POINT2PLANE, plane_flags=1), dict(estimator=capi.EST_PLANE, plane_flags=4), dict(estimator=3)):
        with pytest.raises(capi.Slam3dError):
            capi.IcpHandle(capi.default_params(intr, **bad))
    with capi.IcpHandle(capi.default_params(intr)) as h:
        with pytest.raises(capi.Slam3dError):
            h.set_seg_params(h.seg_params())
        with pytest.raises(capi.Slam3dError):
            h.get_frame_planes(0)

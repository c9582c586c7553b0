"""HIP path vs CPU oracle, through the C-ABI (include/slam3d_icp.h).

Bar (BASELINE.json north_star): correspondence indices bit-exact; pose within 1e-4 rad / 1e-4 m.
The implementation is designed to be bit-identical in T as well; the tests
assert the contractual tolerance and additionally report/guard the stronger property.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from slam3d_gx_amd import capi, synth

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-4   # rad
TRANS_TOL = 1e-4  # m


def _pair(seed, w, h, **kw):
    pr = synth.make_pair(seed, w, h, **kw)
    s4 = synth.backproject_numpy(pr.depth_src, pr.intr)
    t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
    return pr, s4, t4


def _run_both(pr, s4, t4, estimator, iterations, nn_method=1, **kw):
    po = O.params(pr.intr, estimator=estimator, iterations=iterations, nn_method=nn_method, **kw)
    ro = O.icp(s4, t4, po)
    pg = capi.default_params(pr.intr, estimator=estimator, iterations=iterations, max_batch=1, **kw)
    with capi.IcpHandle(pg) as h:
        rg = h.align(s4, t4)
        idx, d2 = h.get_correspondences(0)
        Tt, St = h.get_trace(0)
        _, _, nrm = h.get_clouds(0, normals=(estimator == 0))
    return ro, rg, idx, d2, Tt, St, nrm


@pytest.mark.parametrize("estimator", [0, 1])
@pytest.mark.parametrize("size", [(160, 120), (320, 240)])
def test_small_vs_bruteforce_oracle(gpu_lib, estimator, size):
    pr, s4, t4 = _pair(1000, *size)
    ro, rg, idx, d2, Tt, St, nrm = _run_both(pr, s4, t4, estimator, 5, nn_method=0)
    assert rg["n_src"] == ro["n_src"] and rg["n_tgt"] == ro["n_tgt"]
    assert np.array_equal(idx, ro["idx"]), f"{(idx != ro['idx']).sum()} index mismatches"
    assert np.array_equal(d2, ro["d2"])
    rot, tr = O.pose_error(ro["T_trace"][-1], Tt[-1])
    assert rot <= ROT_TOL and tr <= TRANS_TOL
    assert rg["inliers"] == ro["inliers"] and rg["status"] == ro["status"]
    assert abs(rg["norm"] - ro["norm"]) < 1e-9
    # stronger, by design: every iterate and every 29-sum block identical
    assert np.array_equal(St, ro["sums_trace"])
    assert np.array_equal(Tt, ro["T_trace"])


def test_normals_bitexact(gpu_lib):
    pr, s4, t4 = _pair(1001, 320, 240)
    po = O.params(pr.intr)
    n_or = O.normals(t4, po)
    pg = capi.default_params(pr.intr, iterations=1)
    with capi.IcpHandle(pg) as h:
        h.align(s4, t4)
        _, _, n_gpu = h.get_clouds(0, normals=True)
    assert np.array_equal(n_gpu[..., 3], n_or[..., 3])
    assert np.array_equal(n_gpu, n_or)


def test_backproject_bitexact(gpu_lib):
    pr = synth.make_pair(1002, 640, 480)
    po = O.params(pr.intr)
    ref = O.backproject(pr.depth_src, po)
    with capi.IcpHandle(capi.default_params(pr.intr)) as h:
        got = h.backproject_u16(pr.depth_src)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(np.nan_to_num(got), np.nan_to_num(ref))


@pytest.mark.parametrize("estimator", [0, 1])
def test_full_640x480_config2(gpu_lib, estimator):
    """BASELINE config 2: single 640x480 pair, 20 iterations (oracle NN via exact kd-tree)."""
    pr, s4, t4 = _pair(1000, 640, 480)
    ro, rg, idx, d2, Tt, St, nrm = _run_both(pr, s4, t4, estimator, 20, nn_method=1)
    assert np.array_equal(idx, ro["idx"]), f"{(idx != ro['idx']).sum()} index mismatches"
    assert np.array_equal(d2, ro["d2"])
    rot, tr = O.pose_error(ro["T_trace"][-1], Tt[-1])
    assert rot <= ROT_TOL and tr <= TRANS_TOL
    assert rg["inliers"] == ro["inliers"] and rg["status"] == ro["status"] == 0
    assert np.array_equal(Tt, ro["T_trace"])
    if estimator == 0:   # converges to the analytic pose within the noise floor
        rot_gt, tr_gt = O.pose_error(pr.T_gt, rg["T"])
        assert rot_gt < 2e-3 and tr_gt < 5e-3


@pytest.mark.parametrize("seed", [1000, 1001])
def test_full_640x480_survey_mask(gpu_lib, seed):
    """SURVEY.md 8(d)'s invalid-pixel mask (8x8-pixel Bernoulli holes, p = 0.25: four to five times the hole-border
    length of the default mask) at full size, 20 iterations, as depth images (the projective window search takes part)
    and as clouds: every iterate, the sums and the last indices bit-identical to the oracle (kd-tree)."""
    pr, s4, t4 = _pair(seed, 640, 480, hole_block=8, hole_prob=0.25)
    ro = O.icp(s4, t4, O.params(pr.intr, iterations=20, nn_method=1))
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=20)) as h:
        for depth in (True, False):
            rg = h.align_depth_batch([pr.depth_src], [pr.depth_tgt])[0] if depth else h.align(s4, t4)
            idx, d2 = h.get_correspondences(0)
            Tt, St = h.get_trace(0)
            assert np.array_equal(idx, ro["idx"]), f"depth={depth}: {(idx != ro['idx']).sum()} index mismatches"
            assert np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32))
            assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:20], ro["sums_trace"])
            assert rg["inliers"] == ro["inliers"] and rg["status"] == ro["status"] == 0
    rot_gt, tr_gt = O.pose_error(pr.T_gt, rg["T"])
    assert rot_gt < 3e-3 and tr_gt < 8e-3


def test_batch_equals_single(gpu_lib):
    """Batch mode (config 3 shape, reduced): every pair of a batch equals its single-pair run."""
    seeds = [1000, 1001, 1002, 1003]
    prs = [_pair(s, 320, 240) for s in seeds]
    intr = prs[0][0].intr
    pg = capi.default_params(intr, iterations=6, max_batch=len(seeds))
    with capi.IcpHandle(pg) as h:
        rb = h.align_batch([p[1] for p in prs], [p[2] for p in prs])
        idxb = [h.get_correspondences(b)[0] for b in range(len(seeds))]
    for b, (pr, s4, t4) in enumerate(prs):
        ro = O.icp(s4, t4, O.params(intr, iterations=6, nn_method=1))
        assert np.array_equal(idxb[b], ro["idx"])
        assert np.array_equal(rb[b]["T_raw"], ro["T_trace"][-1])


def test_depth_entry_and_strided_clouds(gpu_lib):
    """align_depth_batch and 32-byte (pcl::PointXYZRGBA-like) records give the same answer."""
    pr, s4, t4 = _pair(1003, 320, 240)
    pg = capi.default_params(pr.intr, iterations=4)
    with capi.IcpHandle(pg) as h:
        r_cloud = h.align(s4, t4)
        r_depth = h.align_depth_batch([pr.depth_src], [pr.depth_tgt])[0]
        s8 = np.zeros(s4.shape[:2] + (8,), dtype=np.float32); s8[..., :3] = s4[..., :3]
        t8 = np.zeros(t4.shape[:2] + (8,), dtype=np.float32); t8[..., :3] = t4[..., :3]
        r_strided = h.align(s8, t8)
    assert np.array_equal(r_cloud["T_raw"], r_depth["T_raw"])
    assert np.array_equal(r_cloud["T_raw"], r_strided["T_raw"])


def test_failure_convention_identity(gpu_lib):
    """Empty target => too few inliers => status 1 and T exactly Identity (src/GraphicEnd.cpp:173,599)."""
    pr, s4, t4 = _pair(1000, 160, 120)
    t_empty = np.full_like(t4, np.nan)
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=3)) as h:
        r = h.align(s4, t_empty)
    ro = O.icp(s4, t_empty, O.params(pr.intr, iterations=3))
    assert r["status"] == ro["status"] == 1
    assert np.array_equal(r["T"], np.eye(4))
    assert r["inliers"] == 0


def test_T_init_and_roundtrip_property(gpu_lib):
    """Size-independent property: aligning (A->B) from identity and (A->B) from the converged pose agree,
    and swapping source/target yields the inverse transform to ICP accuracy."""
    pr, s4, t4 = _pair(1001, 320, 240, noise=False, holes=False)
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=20)) as h:
        r_ab = h.align(s4, t4)
        r_ba = h.align(t4, s4)
        r_warm = h.align(s4, t4, T_init=r_ab["T_raw"])
    E = r_ab["T_raw"] @ r_ba["T_raw"]
    rot, tr = O.pose_error(np.eye(4), E)
    assert rot < 2e-3 and tr < 5e-3
    rot, tr = O.pose_error(r_ab["T_raw"], r_warm["T_raw"])
    assert rot < 1e-4 and tr < 1e-4


@pytest.mark.parametrize("nn_mode", [capi.NN_BRUTE_VALU, capi.NN_BRUTE_MFMA, capi.NN_TILES])
@pytest.mark.parametrize("estimator", [0, 1])
def test_every_nn_mode_is_bit_identical(gpu_lib, nn_mode, estimator):
    """All NN variants (full brute force, tile-pruned) must return the same indices / d2 / poses as the
    brute-force oracle, including with a poor initial guess (large culling radius) and max_corr_dist gates."""
    pr, s4, t4 = _pair(1002, 320, 240)
    for max_corr, T0 in ((0.10, None), (0.03, None), (0.25, synth.pose_from_seed(77, 6.0, 0.10))):
        po = O.params(pr.intr, estimator=estimator, iterations=4, nn_method=0, max_corr_dist=max_corr)
        ro = O.icp(s4, t4, po, T_init=T0)
        pg = capi.default_params(pr.intr, estimator=estimator, iterations=4, nn_mode=nn_mode, max_corr_dist=max_corr)
        with capi.IcpHandle(pg) as h:
            rg = h.align(s4, t4, T_init=T0)
            idx, d2 = h.get_correspondences(0)
            Tt, St = h.get_trace(0)
        assert np.array_equal(idx, ro["idx"]), f"mode {nn_mode} gate {max_corr}: {(idx != ro['idx']).sum()} mismatches"
        assert np.array_equal(d2, ro["d2"])
        assert np.array_equal(St, ro["sums_trace"]) and np.array_equal(Tt, ro["T_trace"])


@pytest.mark.parametrize("bf16", ["1", "0"])
def test_matrix_core_scans_bf16_split_and_f32(gpu_lib, bf16, monkeypatch):
    """NN_BRUTE_MFMA has two forms: k_nn_mfma16 (default: every float split exactly into three bf16 terms, one
    v_mfma_f32_16x16x32_bf16 per 16 x 16 tile) and k_nn_mfma (f32 MFMA, SLAM3D_MFMA_BF16=0).  Both are conservative filters
    in front of the canonical fp32 distance, so both must reproduce the brute-force oracle bit for bit: at 640x480 (where
    |p|, |q| reach the far end of the depth range and the filter's error bound is largest), with a tight gate, from a poor
    initial guess, and in a later iteration where the bounds are millimetres."""
    monkeypatch.setenv("SLAM3D_MFMA_BF16", bf16)
    pr, s4, t4 = _pair(1001, 640, 480)
    for max_corr, T0, iters in ((0.10, None, 3), (0.02, synth.pose_from_seed(5, 1.0, 0.02), 2)):
        po = O.params(pr.intr, iterations=iters, nn_method=1, max_corr_dist=max_corr)
        ro = O.icp(s4, t4, po, T_init=T0)
        pg = capi.default_params(pr.intr, iterations=iters, nn_mode=capi.NN_BRUTE_MFMA, max_corr_dist=max_corr)
        with capi.IcpHandle(pg) as h:
            h.align(s4, t4, T_init=T0)
            idx, d2 = h.get_correspondences(0)
            Tt, St = h.get_trace(0)
        assert np.array_equal(idx, ro["idx"]), f"bf16={bf16} gate {max_corr}: {(idx != ro['idx']).sum()} mismatches"
        assert np.array_equal(d2, ro["d2"])
        assert np.array_equal(St, ro["sums_trace"]) and np.array_equal(Tt, ro["T_trace"])


@pytest.mark.parametrize("variant", ["mfma_bf16", "mfma_f32", "valu", "valu_filter"])
def test_full_scan_kernels_on_a_batch_of_pairs(gpu_lib, variant, monkeypatch):
    """The full-scan kernels index their fragment arrays, bounds and thresholds by pair: a batch of three different pairs in one
    launch sequence must give every pair the result it gets alone in the default (tile-pruned) mode -- indices, d2 and pose bits."""
    env = {"mfma_f32": ("SLAM3D_MFMA_BF16", "0"), "valu_filter": ("SLAM3D_VALU_FILTER", "1")}.get(variant)
    if env:
        monkeypatch.setenv(*env)
    mode = capi.NN_BRUTE_MFMA if variant.startswith("mfma") else capi.NN_BRUTE_VALU
    prs = [_pair(s, 256, 192) for s in (1100, 1101, 1102)]
    intr = prs[0][0].intr
    with capi.IcpHandle(capi.default_params(intr, iterations=4, max_batch=3, nn_mode=mode)) as h:
        rb = h.align_batch([p[1] for p in prs], [p[2] for p in prs])
        got = [h.get_correspondences(b) for b in range(3)]
    for b, (pr, s4, t4) in enumerate(prs):
        with capi.IcpHandle(capi.default_params(intr, iterations=4)) as h1:
            r1 = h1.align(s4, t4)
            idx1, d21 = h1.get_correspondences(0)
        assert np.array_equal(got[b][0], idx1) and np.array_equal(got[b][1], d21), (variant, b)
        assert np.array_equal(rb[b]["T_raw"], r1["T_raw"]) and rb[b]["inliers"] == r1["inliers"]


def test_bf16_scan_at_1280x960_equals_the_tile_search(gpu_lib):
    """BASELINE config 5's size through the bf16 matrix-core scan (1.2 M points: the fragment arrays, the slice count and the
    padding behind the last group at their largest): same correspondences, d2 and pose bits as the tile-pruned search."""
    pr, s4, t4 = _pair(1200, 1280, 960)
    out = []
    for mode in (capi.NN_BRUTE_MFMA, capi.NN_TILES):
        with capi.IcpHandle(capi.default_params(pr.intr, iterations=2, nn_mode=mode)) as h:
            r = h.align(s4, t4)
            idx, d2 = h.get_correspondences(0)
        out.append((idx, d2, r["T_raw"], r["inliers"]))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3]


def test_valu_scan_with_the_expanded_form_filter(gpu_lib, monkeypatch):
    """SLAM3D_VALU_FILTER=1: k_nn_valu takes its chunk minima over the expanded form |q|^2 - 2 p.q (the matrix-core kernels'
    contraction and eps, on the VALU) instead of the canonical distances; flagged chunks are rescanned canonically, so the
    result must be the brute-force oracle's, bit for bit -- 640x480, wide and tight gates."""
    monkeypatch.setenv("SLAM3D_VALU_FILTER", "1")
    pr, s4, t4 = _pair(1003, 640, 480)
    for max_corr, T0, iters in ((0.10, None, 3), (0.02, synth.pose_from_seed(6, 1.0, 0.02), 2)):
        ro = O.icp(s4, t4, O.params(pr.intr, iterations=iters, nn_method=1, max_corr_dist=max_corr), T_init=T0)
        with capi.IcpHandle(capi.default_params(pr.intr, iterations=iters, nn_mode=capi.NN_BRUTE_VALU, max_corr_dist=max_corr)) as h:
            h.align(s4, t4, T_init=T0)
            idx, d2 = h.get_correspondences(0)
            Tt, St = h.get_trace(0)
        assert np.array_equal(idx, ro["idx"]) and np.array_equal(d2, ro["d2"])
        assert np.array_equal(St, ro["sums_trace"]) and np.array_equal(Tt, ro["T_trace"])


def test_bf16_matrix_core_accumulation_error_is_inside_the_filter_bound(gpu_lib, tmp_path):
    """k_nn_mfma16's eps assumes that v_mfma_f32_16x16x32_bf16 returns the exact sum of its (exact) bf16 x bf16 products up to
    24 x 2^-23 of sum |terms| -- a model bound (23 truncating additions), not a measurement.  tools/ubench_bf16acc.hip measures
    what the hardware really does on operands with the kernel's slot pattern and magnitudes (large terms that cancel): the
    worst error must stay inside the bound (it exits non-zero otherwise) and, as a tripwire for a changed matrix core, below
    8 x 2^-24 (measured on gfx950: 2.5)."""
    import re, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "ubench_bf16acc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-w", os.path.join(root, "tools", "ubench_bf16acc.hip"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    worst = float(re.search(r"= ([0-9.]+) x 2\^-24", r.stdout).group(1))
    assert worst < 8.0, r.stdout


def test_odd_image_size_not_multiple_of_tile(gpu_lib):
    """Ragged tiles: width/height not multiples of 8 (tile) or 64 (coarse box)."""
    pr, s4, t4 = _pair(1001, 200, 150)
    ro = O.icp(s4, t4, O.params(pr.intr, iterations=3, nn_method=0))
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=3)) as h:
        h.align(s4, t4)
        idx, d2 = h.get_correspondences(0)
        Tt, _ = h.get_trace(0)
    assert np.array_equal(idx, ro["idx"]) and np.array_equal(Tt, ro["T_trace"])


def test_dense_mode_two_emulated_ranks(gpu_lib):
    """Dense mode (config 5 shape, reduced): one pair, source rows sharded over 2 'ranks' (two handles on one
    GPU, host-side integer sum standing in for the RCCL all-reduce).  Indices are unaffected by the sharding, and
    the int64 fixed-point sums make the pose independent of it too."""
    from slam3d_gx_amd import dense, shard
    pr, s4, t4 = _pair(1002, 320, 240)
    iters = 6
    ro = O.icp(s4, t4, O.params(pr.intr, iterations=iters, nn_method=1))
    hs = [capi.IcpHandle(capi.default_params(pr.intr, iterations=iters)) for _ in range(2)]
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
        idx = [h.get_correspondences(0)[0] for h in hs]
    finally:
        for h in hs:
            h.close()
    assert np.array_equal(res[0]["T_raw"], res[1]["T_raw"])            # every rank holds the same pose
    rot, tr = O.pose_error(ro["T_trace"][-1], res[0]["T_raw"])
    assert rot <= ROT_TOL and tr <= TRANS_TOL
    # integer fixed-point sums are order-free: the sharded run equals the unsharded oracle bit for bit
    assert np.array_equal(res[0]["T_raw"], ro["T_trace"][-1])
    assert res[0]["inliers"] == ro["inliers"]
    r0, r1 = shard.dense_row_range(pr.intr.height, 2, 0)
    W = pr.intr.width
    merged = np.where(np.arange(idx[0].size) < r1 * W, idx[0], idx[1])
    assert np.array_equal(merged, ro["idx"])
    # single-rank dense run == batch run, bit for bit
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=iters)) as h:
        h.set_clouds_host(0, s4, t4)
        r1rank = dense.dense_align(h, 1, 0)
    assert np.array_equal(r1rank["T_raw"], ro["T_trace"][-1])
    # device-resident exchange buffer (what bench.py --mode dense runs over RCCL): same bits again
    import torch
    d_sums = torch.zeros(36, dtype=torch.int64, device="cuda:0")
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=iters)) as h:
        h.set_clouds_host(0, s4, t4)
        rdev = dense.dense_align_device(h, 1, 0, d_sums, None, torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(rdev["T_raw"], r1rank["T_raw"]) and rdev["inliers"] == r1rank["inliers"]
    assert rdev["rmse"] == r1rank["rmse"]


def test_config3_batch_of_full_size_pairs(gpu_lib):
    """BASELINE config 3 shape (16 of the 64 pairs to keep the oracle leg short): one batched call on 640x480 pairs.
    Every pair converges to its analytic pose within the noise floor; sampled pairs equal the oracle bit for bit."""
    B = 16
    prs = [_pair(1000 + i, 640, 480) for i in range(B)]
    intr = prs[0][0].intr
    with capi.IcpHandle(capi.default_params(intr, iterations=20, max_batch=B)) as h:
        res = h.align_batch([p[1] for p in prs], [p[2] for p in prs])
        idx_first, idx_last = h.get_correspondences(0)[0], h.get_correspondences(B - 1)[0]
    assert all(r["status"] == 0 for r in res)
    for (pr, _, _), r in zip(prs, res):
        rot, tr = O.pose_error(pr.T_gt, r["T"])
        assert rot < 3e-3 and tr < 1e-2, (pr.seed, rot, tr)
    for b, idx in ((0, idx_first), (B - 1, idx_last)):
        pr, s4, t4 = prs[b]
        ro = O.icp(s4, t4, O.params(intr, iterations=20, nn_method=1))
        assert np.array_equal(idx, ro["idx"])
        assert np.array_equal(res[b]["T_raw"], ro["T_trace"][-1])


def test_config5_dense_1280x960(gpu_lib):
    """BASELINE config 5 shape: one 1280x960 pair (1.2 M points), point-to-plane, vs the kd-tree oracle."""
    pr, s4, t4 = _pair(2000, 1280, 960)
    iters = 8
    ro = O.icp(s4, t4, O.params(pr.intr, iterations=iters, nn_method=1))
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=iters)) as h:
        r = h.align(s4, t4)
        idx, d2 = h.get_correspondences(0)
    assert r["n_src"] == ro["n_src"] and r["n_tgt"] == ro["n_tgt"]
    assert np.array_equal(idx, ro["idx"]) and np.array_equal(d2, ro["d2"])
    rot, tr = O.pose_error(ro["T_trace"][-1], r["T_raw"])
    assert rot <= ROT_TOL and tr <= TRANS_TOL
    assert np.array_equal(r["T_raw"], ro["T_trace"][-1])


def test_fit_planes_vs_oracle(gpu_lib):
    """Row a6: per-plane {sum p, sum pp^T, n} -> eigen -> (n, d) with d >= 0 (src/GraphicEnd.cpp:360-387).
    Integer fixed-point moments (exact, order-free) + the spec's Jacobi on the device: bit-identical to the oracle."""
    pr = synth.make_pair(1001, 640, 480)
    d, lab = synth.render_depth(np.eye(4), pr.intr, 1001, 1, hole_block=32, want_labels=True)
    assert np.array_equal(d, pr.depth_src)
    cloud = synth.backproject_numpy(d, pr.intr)
    ref_planes, ref_counts = O.fit_planes(cloud, lab, 3)
    with capi.IcpHandle(capi.default_params(pr.intr)) as h:
        got = h.fit_planes(cloud, lab, 3)
        # also through 32-byte PointXYZRGBA-like records
        c8 = np.zeros(cloud.shape[:2] + (8,), dtype=np.float32); c8[..., :3] = cloud[..., :3]
        got8 = h.fit_planes(c8, lab, 3)
    for k in range(3):
        assert got[k]["count"] == ref_counts[k] == int((lab == k).sum())
        assert np.array_equal(got[k]["coeff"], ref_planes[k])
        assert got[k]["coeff"][3] >= 0 and abs(np.linalg.norm(got[k]["coeff"][:3]) - 1) < 1e-6
        assert np.array_equal(got[k]["coeff"], got8[k]["coeff"])
    expect = np.array([[0, -1, 0, 1.2], [1, 0, 0, 2.0], [0, 0, -1, 4.5]])
    assert np.allclose(np.stack([g["coeff"] for g in got]), expect, atol=5e-3)


def test_profiling_is_opt_in_and_does_not_change_results(gpu_lib):
    """Per-launch events (slam3d_icp_set_profiling) only add timing: same T, and the iteration timings are
    refused for a run that was not profiled."""
    pr, s4, t4 = _pair(31, 320, 240)
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=6, max_batch=1)) as h:
        r0 = h.align(s4, t4)
        with pytest.raises(capi.Slam3dError) as e:
            h.get_iteration_timings()
        assert e.value.code == -5
        tm = h.get_timings()
        assert tm["nn_ms"] == 0.0 and tm["total_ms"] > 0.0
        h.set_profiling(True)
        r1 = h.align(s4, t4)
        ms = h.get_iteration_timings()
        assert ms.shape == (6,) and (ms > 0).all()
        assert h.get_timings()["nn_ms"] > 0.0
    assert np.array_equal(r0["T"], r1["T"]) and r0["inliers"] == r1["inliers"]


@pytest.mark.parametrize("size,seed", [((160, 120), 3), ((320, 240), 1), ((640, 480), 7)])
def test_plane_segmentation_matches_oracle_bit_for_bit(gpu_lib, size, seed):
    """f-2: batched RANSAC plane segmentation.  Integer consensus counts, integer fixed-point moments and the
    spec's Jacobi solve leave no room for rounding differences: labels, coefficients, centroids and counts
    are identical to oracle/seg_oracle.c."""
    pr, s4, _ = _pair(1000 + seed, *size)
    po, lo = O.segment_planes(s4, seed=seed)
    with capi.IcpHandle(capi.default_params(pr.intr, max_batch=1)) as h:
        pg, lg = h.segment_planes(s4, h.seg_params(seed=seed))
    assert len(pg) == len(po) and len(po) >= 2
    assert np.array_equal(lg, lo)
    for a, b in zip(pg, po):
        assert a["count"] == b["count"]
        assert np.array_equal(a["coeff"], b["coeff"]) and np.array_equal(a["centroid"], b["centroid"])
        assert a["coeff"][3] >= 0 and abs(np.linalg.norm(a["coeff"][:3]) - 1) < 1e-6     # src/GraphicEnd.cpp:383-387


def test_plane_segmentation_batch_on_device_and_edge_cases(gpu_lib):
    import torch
    pr, s4, t4 = _pair(1003, 320, 240)
    N = 320 * 240
    empty = np.full((N, 4), np.nan, dtype=np.float32)                      # no valid point: zero planes
    few = empty.copy(); few[:2] = [[0, 0, 1, 1], [0.1, 0, 1, 1]]          # two points: no plane possible
    clouds = [np.ascontiguousarray(c, dtype=np.float32).reshape(N, 4) for c in (s4, t4, empty, few)]
    d = torch.from_numpy(np.stack(clouds)).to("cuda:0")
    d_lab = torch.zeros((4, N), dtype=torch.int32, device="cuda:0")
    with capi.IcpHandle(capi.default_params(pr.intr, max_batch=4)) as h:
        sp = h.seg_params(seed=11, max_planes=4, hypotheses=48)
        out = h.segment_planes_device([d.data_ptr() + i * N * 16 for i in range(4)], sp, d_lab.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
    lab = d_lab.cpu().numpy()
    for i, c in enumerate(clouds):
        po, lo = O.segment_planes(c, seed=11, max_planes=4, hypotheses=48)
        assert len(out[i]) == len(po)
        assert np.array_equal(lab[i], lo)
        for a, b in zip(out[i], po):
            assert a["count"] == b["count"] and np.array_equal(a["coeff"], b["coeff"])
    assert len(out[2]) == 0 and len(out[3]) == 0 and (lab[2] == -2).all() and (lab[3][:2] == -1).all()


def test_new_entry_points_reject_bad_arguments(gpu_lib):
    """f-1 / f-2 / dense-device entry points: usage errors are SLAM3D_E_INVALID / _STATE, never a crash."""
    import ctypes as C
    pr, s4, _ = _pair(1000, 160, 120)
    N = 160 * 120
    with capi.IcpHandle(capi.default_params(pr.intr, max_batch=1)) as h:
        lib, hp = h.lib, h._h
        sp = h.seg_params()
        planes = (capi.Plane * 8)()
        npl = (C.c_int32 * 1)()
        m = C.c_int32(0)
        buf = np.zeros((N, 4), np.float32)
        assert lib.slam3d_voxel_grid(hp, None, 10, C.c_float(0.03), capi._vp(buf), C.byref(m)) == -1
        assert lib.slam3d_voxel_grid(hp, capi._vp(buf), N + 1, C.c_float(0.03), capi._vp(buf), C.byref(m)) == -1     # larger than the handle
        assert lib.slam3d_voxel_grid(hp, capi._vp(buf), 10, C.c_float(0.0), capi._vp(buf), C.byref(m)) == -1         # leaf must be > 0
        assert lib.slam3d_voxel_grid(hp, capi._vp(buf), 0, C.c_float(0.03), capi._vp(buf), C.byref(m)) == 0 and m.value == 0
        for bad in (dict(max_planes=0), dict(max_planes=9), dict(hypotheses=0), dict(hypotheses=65), dict(distance_threshold=0.0)):
            with pytest.raises(capi.Slam3dError) as e:
                h.segment_planes(s4, h.seg_params(**bad))
            assert e.value.code == -1
        assert lib.slam3d_segment_planes(hp, None, C.byref(sp), planes, npl, None) == -1
        assert lib.slam3d_segment_planes_device(hp, 2, None, C.byref(sp), planes, npl, None, None) == -1             # B > max_batch / no clouds
        # dense device calls before dense_begin
        assert lib.slam3d_icp_dense_partial_device(hp, C.c_void_p(8), None) == -5
        assert lib.slam3d_icp_dense_update_device(hp, None, None) == -1
        assert lib.slam3d_icp_set_profiling(None, 1) == -1


def test_batches_larger_than_one_argument_block(gpu_lib):
    """The slot / pointer tables travel as kernel arguments in blocks of 32: a batch of 40 small pairs (and 40 frames
    for the segmentation) must give, pair by pair, what single calls give."""
    import torch
    B, W, H = 40, 96, 72
    prs = [synth.make_pair(3000 + i, W, H) for i in range(B)]
    src = [synth.backproject_numpy(p.depth_src, p.intr) for p in prs]
    tgt = [synth.backproject_numpy(p.depth_tgt, p.intr) for p in prs]
    Ti = [np.eye(4) for _ in range(B)]
    Ti[37] = synth.pose_from_seed(9, max_angle_deg=0.5, max_trans=0.01)
    with capi.IcpHandle(capi.default_params(prs[0].intr, iterations=4, max_batch=B)) as h:
        res = h.align_batch(src, tgt, Ti)
        idx37, _ = h.get_correspondences(37)
        d = torch.from_numpy(np.stack([s.reshape(-1, 4) for s in src])).to("cuda:0")
        planes = h.segment_planes_device([d.data_ptr() + i * W * H * 16 for i in range(B)], h.seg_params(seed=5))
    for i in (0, 31, 32, 37, 39):
        ro = O.icp(src[i], tgt[i], O.params(prs[i].intr, iterations=4, nn_method=0), T_init=Ti[i])
        assert np.array_equal(res[i]["T_raw"], ro["T_trace"][-1]) and res[i]["inliers"] == ro["inliers"]
        po, _ = O.segment_planes(src[i], seed=5)
        assert len(planes[i]) == len(po) and all(np.array_equal(a["coeff"], b["coeff"]) for a, b in zip(planes[i], po))
    assert np.array_equal(idx37, O.icp(src[37], tgt[37], O.params(prs[37].intr, iterations=4, nn_method=0), T_init=Ti[37])["idx"])


@pytest.mark.parametrize("case_seed,force_throughput_build", [(11, False), (12, False), (13, True), (14, True)])
def test_randomised_configurations_stay_bit_identical(gpu_lib, case_seed, force_throughput_build, monkeypatch):
    """A slice of tools/soak_parity.py (300 random cases run clean on the MI355X): random size, gate, estimator,
    iteration count, initial guess and sparsity; indices, d2 bits, every iterate and the sums equal the oracle's."""
    if force_throughput_build:      # the non-cooperative <3, 8> build normally serves launches of >= 8 pairs
        monkeypatch.setenv("SLAM3D_DENSE_BATCH", "1")
    rng = np.random.default_rng(case_seed)
    for _ in range(4):
        W = int(rng.choice([64, 104, 160, 200])); H = int(rng.choice([48, 72, 120, 150]))
        seed = int(rng.integers(0, 1 << 30)); est = int(rng.integers(0, 2)); iters = int(rng.integers(1, 6))
        gate = float(rng.choice([0.01, 0.03, 0.1, 0.3, 1.0]))
        pr = synth.make_pair(seed, W, H, noise=bool(rng.integers(0, 2)), holes=bool(rng.integers(0, 2)))
        s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
        mode, Ti = int(rng.integers(0, 4)), None
        if mode == 1:
            Ti = synth.pose_from_seed(seed + 1, max_angle_deg=8.0, max_trans=0.3)
        elif mode == 2:
            s4 = s4.copy(); s4[rng.random(s4.shape[:2]) < 0.7] = np.nan
        elif mode == 3:
            t4 = t4.copy(); t4[rng.random(t4.shape[:2]) < 0.9] = np.nan
        kw = dict(estimator=est, iterations=iters, max_corr_dist=gate)
        ro = O.icp(s4, t4, O.params(pr.intr, nn_method=0, **kw), T_init=Ti)
        with capi.IcpHandle(capi.default_params(pr.intr, max_batch=1, **kw)) as h:
            rg = h.align(s4, t4, Ti)
            idx, d2 = h.get_correspondences(0)
            Tt, St = h.get_trace(0)
        ctx = dict(W=W, H=H, seed=seed, est=est, iters=iters, gate=gate, mode=mode)
        assert np.array_equal(idx, ro["idx"]), ctx
        assert np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32)), ctx
        assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:iters], ro["sums_trace"]), ctx
        assert rg["status"] == ro["status"] and rg["inliers"] == ro["inliers"], ctx


def test_two_handles_in_flight_give_the_sequential_results(gpu_lib):
    """What bench.py's double buffering relies on: two handles (one HIP stream each) with runs queued back to back;
    fetch_results waits for its own run only and returns exactly what a lone run returns."""
    import torch
    prs = [synth.make_pair(5000 + i, 320, 240) for i in range(2)]
    src = [torch.from_numpy(synth.backproject_numpy(p.depth_src, p.intr)).to("cuda:0") for p in prs]
    tgt = [torch.from_numpy(synth.backproject_numpy(p.depth_tgt, p.intr)).to("cuda:0") for p in prs]
    hs = [capi.IcpHandle(capi.default_params(prs[0].intr, iterations=8, max_batch=1)) for _ in range(2)]
    try:
        lone = []
        for k in range(2):
            hs[k].set_clouds_device(0, src[k].data_ptr(), tgt[k].data_ptr())
            hs[k].run(1)
            lone.append(hs[k].fetch_results(1)[0])
        for rep in range(5):                       # queue A, queue B, then fetch A, fetch B (graph replays on two streams)
            hs[0].run(1); hs[1].run(1)
            both = [hs[0].fetch_results(1)[0], hs[1].fetch_results(1)[0]]
            for k in range(2):
                assert np.array_equal(both[k]["T_raw"], lone[k]["T_raw"]) and both[k]["inliers"] == lone[k]["inliers"]
    finally:
        for h in hs:
            h.close()
    for k in range(2):
        ro = O.icp(src[k].cpu().numpy(), tgt[k].cpu().numpy(), O.params(prs[k].intr, iterations=8, nn_method=1))
        assert np.array_equal(lone[k]["T_raw"], ro["T_trace"][-1])


def test_bench_two_ranks_on_one_device(gpu_lib):
    """bench.py's N>1 code path (pairs per rank, per-step pose gather, max over ranks) with two ranks sharing this GPU
    over gloo: RCCL needs one GPU per rank, everything else is the code the driver runs.  (The dense N>1 form needs
    RCCL; its loop is covered by test_frames_comm.py with a one-rank communicator and by tests/test_shard_gloo.py.)"""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--width", "320", "--height", "240", "--iterations", "6", "--no-cpu-baseline", "--no-bruteforce",
           "--pairs-per-step", "6", "--pool", "4", "--profile-aligns", "4", "--dist-backend", "gloo", "--one-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["status"][0] == 0 and d["scaling"] == "weak"
    assert d["config"]["gathered_pose_records"] == 12 and d["config"]["pairs_per_step_per_gpu"] == 6


def test_bench_gpus2_launched_as_a_plain_process(gpu_lib):
    """`python bench.py --gpus 2 ...` with NO launcher must start its own ranks (it re-executes itself
    through torch.distributed.run on 127.0.0.1) instead of exiting non-zero; two ranks on this one GPU over gloo."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--width", "320", "--height", "240", "--iterations", "6", "--no-cpu-baseline", "--no-bruteforce",
           "--pairs-per-step", "6", "--pool", "4", "--profile-aligns", "4", "--overlap-aligns", "0", "--dist-backend", "gloo", "--one-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["status"][0] == 0
    assert d["survey_8d"]["gpus"] == 2 and d["survey_8d"]["icp_iters_per_s"] == d["value"]
    assert d["rccl_ranks"] == 0 and d["pose_exchange"].startswith("gloo")


@pytest.mark.parametrize("mode", ["0", "1", "2", "3"])
def test_head_solve_modes_are_bit_identical(gpu_lib, mode, monkeypatch):
    """The solve at the head of the next NN launch (default, 1), the same with every block solving for itself instead of
    polling block 0 (2: the fallback path of a poller that gives up) and the two-launch form (0): every iterate, every
    sum, the flags and the result record equal the oracle's -- single pair, a 3-pair launch, and a solve that fails."""
    monkeypatch.setenv("SLAM3D_HEAD_SOLVE", mode)
    iters = 9
    prs = [_pair(3000 + i, 320, 240) for i in range(3)]
    intr = prs[0][0].intr
    with capi.IcpHandle(capi.default_params(intr, iterations=iters, max_batch=3)) as h:
        for rep in range(2):                                          # second time: graph replay on cached frames
            res = h.align_batch([p[1] for p in prs], [p[2] for p in prs])
            for b, (pr, s4, t4) in enumerate(prs):
                ro = O.icp(s4, t4, O.params(intr, iterations=iters, nn_method=1))
                Tt, St = h.get_trace(b)
                assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:iters], ro["sums_trace"]), (mode, rep, b)
                assert np.array_equal(res[b]["T_raw"], ro["T_trace"][-1]) and res[b]["inliers"] == ro["inliers"] and res[b]["status"] == ro["status"]
                assert np.array_equal(h.get_correspondences(b)[0], ro["idx"])
        one = h.align(prs[1][1], prs[1][2])                           # B = 1 on the same handle (another graph)
        assert np.array_equal(one["T_raw"], res[1]["T_raw"])
    w, hh = 160, 120
    intr = synth.Intrinsics.scaled(w, hh)
    wall = synth.backproject_numpy(np.full((hh, w), 2000, dtype=np.uint16), intr)
    Ti = np.eye(4); Ti[0, 3] = 0.01
    with capi.IcpHandle(capi.default_params(intr, iterations=3)) as hd:
        r = hd.align(wall, wall, Ti)
        Tt, St = hd.get_trace(0)
    ro = O.icp(wall, wall, O.params(intr, iterations=3, nn_method=0), T_init=Ti)
    assert r["status"] == ro["status"] == 3 and np.array_equal(r["T_raw"], ro["T_trace"][-1])
    assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:3], ro["sums_trace"])


@pytest.mark.parametrize("cert", ["1", "0"])
def test_clearance_certificates_change_no_bit(gpu_lib, cert, monkeypatch):
    """Clearance certificates (icp_kernels.hpp: from the seventh iteration on a lane whose nearest neighbour provably did not
    change is not searched again) on and off: every iterate, every sum, the indices and d2 of the last iteration equal the
    oracle's over LONG runs (40 iterations: certificate chains of 30+ skipped searches), with exact ties (duplicated target
    points: clearance 0, never certified), a tight and a wide gate, both estimators, a batch, and a poor initial guess."""
    monkeypatch.setenv("SLAM3D_CERT", cert)
    cases = [dict(seed=1000, w=320, h=240, est=0, iters=40, gate=0.10), dict(seed=1004, w=320, h=240, est=1, iters=40, gate=0.10),
             dict(seed=1005, w=160, h=120, est=0, iters=40, gate=0.02), dict(seed=1006, w=320, h=240, est=0, iters=30, gate=0.5)]
    for cse in cases:
        pr, s4, t4 = _pair(cse["seed"], cse["w"], cse["h"])
        t4 = t4.copy()
        t4[:, 1::7, :] = t4[:, 0::7, :][:, : t4[:, 1::7, :].shape[1], :]        # every seventh column duplicates its neighbour: exact ties
        Ti = synth.pose_from_seed(cse["seed"] + 7, 1.0, 0.02) if cse["seed"] % 2 else None
        po = O.params(pr.intr, estimator=cse["est"], iterations=cse["iters"], nn_method=1, max_corr_dist=cse["gate"])
        ro = O.icp(s4, t4, po, T_init=Ti)
        with capi.IcpHandle(capi.default_params(pr.intr, estimator=cse["est"], iterations=cse["iters"], max_corr_dist=cse["gate"])) as h:
            for rep in range(2):
                rg = h.align(s4, t4, Ti)
                idx, d2 = h.get_correspondences(0)
                Tt, St = h.get_trace(0)
                assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[: cse["iters"]], ro["sums_trace"]), (cse, rep)
                assert np.array_equal(idx, ro["idx"]) and np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32)), (cse, rep)
                assert rg["inliers"] == ro["inliers"] and rg["status"] == ro["status"]
    prs = [_pair(3200 + i, 320, 240) for i in range(3)]
    with capi.IcpHandle(capi.default_params(prs[0][0].intr, iterations=25, max_batch=3)) as h:
        res = h.align_batch([p[1] for p in prs], [p[2] for p in prs])
        for b, (pr, s4, t4) in enumerate(prs):
            ro = O.icp(s4, t4, O.params(pr.intr, iterations=25, nn_method=1))
            assert np.array_equal(h.get_trace(b)[0].reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(h.get_correspondences(b)[0], ro["idx"]), b


def test_launch_stamps_are_ordered_and_change_nothing(gpu_lib):
    """slam3d_icp_set_stamping: every NN / solve launch reports (start, end) on the device's real-time counter; the
    launches of a run follow each other, two handles in flight share the clock, and the results are bit-identical with
    the stamps on and off."""
    pr, s4, t4 = _pair(1000, 320, 240)
    iters = 6
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=iters)) as h, capi.IcpHandle(capi.default_params(pr.intr, iterations=iters)) as h2:
        want = h.align(s4, t4)
        with pytest.raises(capi.Slam3dError):
            h.get_stamps()                                            # no ring yet
        h.set_stamping(3); h2.set_stamping(3)
        h.set_clouds_host(0, s4, t4); h2.set_clouds_host(0, s4, t4)
        for _ in range(5):                                            # five runs through a ring of three
            h.run(1); h2.run(1)
            got, got2 = h.fetch_results(1)[0], h2.fetch_results(1)[0]
        ring, ring2 = h.get_stamps().astype(np.int64), h2.get_stamps().astype(np.int64)
        assert ring.shape == (3, 2 * iters, 2) and ring2.shape == (3, 2 * iters, 2)
        assert (ring[1:, 0, 0] > ring[:-1, iters - 1, 1]).all()       # oldest first: run k+1 starts after run k's last NN launch ended
        st, st2 = ring[-1], ring2[-1]
        h.set_stamping(0)
        again = h.align(s4, t4)
    for g in (got, got2, again):
        assert np.array_equal(g["T_raw"], want["T_raw"]) and g["inliers"] == want["inliers"]
    for s in (st, st2):
        nn = s[:iters]
        assert (nn[:, 0] > 0).all() and (nn[:, 1] > nn[:, 0]).all()                   # every NN launch ran, end after start
        assert (nn[1:, 0] >= nn[:-1, 1]).all()                                       # iteration k+1 starts after iteration k ended
        dur_us = (nn[:, 1] - nn[:, 0]) * 0.01
        assert (dur_us > 1).all() and (dur_us < 5000).all()
    # one clock for both handles: the two runs were queued back to back, so they lie within a few milliseconds of each other
    assert abs(int(st[0, 0]) - int(st2[0, 0])) * 1e-8 < 0.05


def test_bench_single_gpu_line_has_the_contract_fields(gpu_lib, tmp_path):
    """The default code path of bench.py at a small size: the LAST stdout line is one JSON object with the contract's fields,
    roofline, cpu_baseline and parity, shorter than 6,000 bytes (round 5's 21.5 KB line did not parse on the
    driver's side); everything else -- legs, notes, per-iteration arrays -- is in the legs file it names.  RCCL pose gather
    forced on (one rank)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    legs = str(tmp_path / "legs.json")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--width", "320", "--height", "240",
           "--iterations", "6", "--pairs-per-step", "6", "--pool", "4", "--profile-aligns", "4", "--force-collective", "--in-flight", "4",
           "--legs-file", legs]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    last = out.stdout.rstrip("\n").splitlines()[-1]
    assert last.startswith("{") and len(last.encode()) < 6000, len(last)
    d = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_vs_oracle", "survey_8d"):
        assert k in d, k
    assert d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["value"] > 0 and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0 and 0 < d["roofline"]["frac"] < 1 and "traffic" in d["roofline"]
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-4
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["sample"]
    assert d["parity_vs_oracle"]["idx_mismatches"] == 0 and d["parity_vs_oracle"]["T_bit_identical"]
    assert d["config"]["gathered_pose_records"] == 6 and "workload" in d["config"]
    assert d["rccl_ranks"] == 1 and d["pose_exchange"].startswith("rccl")
    assert d["config"]["coarse_iterations"] == 3 and d["config"]["noise_sigma_over_z2"] == 0.0012 and d["config"]["hole_block_px"] == 8
    # the roofline's algorithmic bytes count the three coarse launches of the six at a quarter of the sources
    assert d["roofline"]["algorithmic_bytes_per_launch"] < 16 * 0.7 * 76800 + 24 * 76800
    # the complete object: same headline, plus what the line leaves out
    full = json.load(open(legs))
    assert abs(full["value"] - d["value"]) <= 1e-5 * d["value"] and os.path.samefile(os.path.join(root, d["legs_file"]), legs)
    assert len(full["per_rank"]) == 1 and full["per_rank"][0]["h2d_GBps"] > 1 and full["per_rank"][0]["gather_us_per_step"] > 0
    ov = full["overlap"]
    assert ov["mean_resident_nn_kernels"] > 0 and ov["nn_launch_us_overlapped"]["mean"] > 0 and ov["in_flight"] == 4
    assert abs(ov["sum_nn_us_per_alignment"] / ov["mean_resident_nn_kernels"] - ov["per_alignment_wall_us_device_clock"]) < 1e-6 * ov["per_alignment_wall_us_device_clock"]
    # every stdout line that looks like JSON parses (the per-leg lines in front of the final one)
    for ln in out.stdout.splitlines():
        if ln.startswith("{"):
            json.loads(ln)


@pytest.mark.parametrize("estimator", [0, 1])
def test_zero_distances_and_duplicate_targets(gpu_lib, estimator):
    """d2 == 0 makes the packed key a DENORMAL double (high word 0, mantissa = pixel index): the v_min_f64 key minimum
    must still order by pixel.  Source == target gives d2 = 0 everywhere; duplicated target points give exact ties
    that must resolve to the smallest pixel index, like the oracle's ascending scan."""
    pr, s4, _ = _pair(77, 160, 120, noise=False, holes=False)
    t4 = s4.copy()
    # duplicate every second column into its left neighbour: exact ties at d2 = 0 between pixel u-1 and u
    t4[:, 1::2, :] = t4[:, 0::2, :]
    for src, tgt in ((s4, s4), (s4, t4)):
        ro = O.icp(src, tgt, O.params(pr.intr, estimator=estimator, iterations=1, nn_method=0))
        for env in (None, "1"):
            import os
            if env:
                os.environ["SLAM3D_DENSE_BATCH"] = env
            try:
                with capi.IcpHandle(capi.default_params(pr.intr, estimator=estimator, iterations=1, max_batch=1)) as h:
                    h.align(src, tgt)
                    idx, d2 = h.get_correspondences(0)
                    Tt, _ = h.get_trace(0)
            finally:
                os.environ.pop("SLAM3D_DENSE_BATCH", None)
            assert np.array_equal(idx, ro["idx"]) and np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32))
            assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"])
    valid = ro["idx"] >= 0
    assert (ro["d2"][valid] == 0).mean() > 0.4            # the duplicated columns really produce zero-distance ties
    u = np.arange(160)[None, :].repeat(120, 0).reshape(-1)
    even = valid & (u % 2 == 0) & (ro["d2"] == 0)       # target pixels u and u+1 both hold source point u: exact tie
    if estimator == 1:                                   # (point-to-plane only matches targets that kept a normal)
        assert even.any() and (ro["idx"][even] == np.nonzero(even)[0]).all()  # ... resolved to the smaller pixel index


@pytest.mark.gpu
@pytest.mark.parametrize("force_throughput_build", [False, True])
def test_tiny_frames_and_ownership_map_reuse(gpu_lib, force_throughput_build, monkeypatch):
    """Frames of one to a few dozen tiles (fewer tiles than XCD bands, bands of zero tiles, grid slack rounding) and
    the tile-ownership map carried from one run to the next on the same handle (k_balance: any assignment must own
    every tile exactly once, so indices, iterates and sums stay the oracle's)."""
    if force_throughput_build:
        monkeypatch.setenv("SLAM3D_DENSE_BATCH", "1")
    for (W, H) in [(8, 8), (16, 8), (24, 16), (40, 24), (72, 40), (136, 8)]:
        prs = [synth.make_pair(2000 + k, W, H, holes=bool(k & 1)) for k in range(3)]
        with capi.IcpHandle(capi.default_params(prs[0].intr, iterations=4, max_corr_dist=0.5)) as h:
            for pr in prs:      # second and third run start from the previous run's ownership map
                s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
                ro = O.icp(s4, t4, O.params(pr.intr, iterations=4, nn_method=0, max_corr_dist=0.5))
                rg = h.align(s4, t4)
                idx, d2 = h.get_correspondences(0)
                Tt, St = h.get_trace(0)
                ctx = dict(W=W, H=H)
                assert np.array_equal(idx, ro["idx"]), ctx
                assert np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32)), ctx
                assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:4], ro["sums_trace"]), ctx
                assert rg["status"] == ro["status"] and rg["inliers"] == ro["inliers"], ctx


@pytest.mark.parametrize("size,estimator,guess", [((640, 480), 0, False), ((640, 480), 1, True), ((200, 150), 0, True), ((104, 72), 0, False)])
@pytest.mark.parametrize("window_search", ["on", "off"])
def test_depth_frames_with_and_without_the_projective_window_search(gpu_lib, size, estimator, guess, window_search, monkeypatch):
    """Frames that enter as depth images are back-projected by the library, so their targets are camera-consistent and the
    lanes whose bound is a few pixels wide find their neighbour in a window around their projection (DESIGN.md section 5;
    the bound itself: tests/test_gap_bound.py).  Same bits as the brute-force oracle with the window search on and off
    (SLAM3D_PROJ_SEARCH=0), per iteration, with holes and with a poor initial guess."""
    if window_search == "off":
        monkeypatch.setenv("SLAM3D_PROJ_SEARCH", "0")
    pr = synth.make_pair(3100 + size[0], *size, holes=True)
    iters = 6
    kw = dict(estimator=estimator, iterations=iters)
    po = O.params(pr.intr, nn_method=0, **kw)
    s4, t4 = O.backproject(pr.depth_src, po), O.backproject(pr.depth_tgt, po)
    Ti = synth.pose_from_seed(5, max_angle_deg=2.0, max_trans=0.04) if guess else None
    ro = O.icp(s4, t4, po, T_init=Ti)
    with capi.IcpHandle(capi.default_params(pr.intr, max_batch=1, **kw)) as h:
        h.set_corr_trace(True)
        rg = h.align_depth_batch([pr.depth_src], [pr.depth_tgt], None if Ti is None else [Ti])[0]
        idx, d2 = h.get_correspondences(0)
        Tt, St = h.get_trace(0)
        mid, mid4 = h.get_correspondences_at(2, 0), h.get_correspondences_at(4, 0)
    assert np.array_equal(idx, ro["idx"]), int((idx != ro["idx"]).sum())
    assert np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32))
    assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:iters], ro["sums_trace"])
    assert rg["inliers"] == ro["inliers"] and rg["status"] == ro["status"]
    # iteration 2 of a run is a coarse one (spec S4c): the sources of every fourth tile, at the trace pose
    want, _, _ = O.nn_once(s4, t4, po, T=ro["T_trace"][2], use_normals=(estimator == 0), coarse=True)
    assert np.array_equal(mid, want) and (mid >= 0).sum() < 0.4 * (idx >= 0).sum()
    r1 = O.icp(s4, t4, O.params(pr.intr, nn_method=0, **{**kw, "iterations": 1}), T_init=ro["T_trace"][4])
    assert np.array_equal(mid4, r1["idx"])


def test_handles_of_different_geometry_side_by_side_and_slot_reuse(gpu_lib):
    """The search kernel reads its per-handle constants (pointers, geometry, tile grid, iteration count) from a table in
    constant memory, one entry per live handle: handles of DIFFERENT sizes / estimators / iteration counts alive together must
    each get their own oracle result, in any order of use, and entries must be recycled (more create / destroy cycles than
    the table has entries)."""
    cfgs = [(64, 48, 0, 3), (104, 72, 1, 5), (160, 120, 0, 8), (40, 24, 0, 2)]
    prs = [synth.make_pair(3100 + k, W, H) for k, (W, H, _, _) in enumerate(cfgs)]
    hs = [capi.IcpHandle(capi.default_params(pr.intr, estimator=e, iterations=it, max_corr_dist=0.3)) for pr, (_, _, e, it) in zip(prs, cfgs)]
    try:
        for k in (2, 0, 3, 1, 0, 2):
            pr, (W, H, e, it) = prs[k], cfgs[k]
            s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
            ro = O.icp(s4, t4, O.params(pr.intr, estimator=e, iterations=it, nn_method=0, max_corr_dist=0.3))
            rg = hs[k].align(s4, t4)
            idx, _ = hs[k].get_correspondences(0)
            Tt, _ = hs[k].get_trace(0)
            assert np.array_equal(idx, ro["idx"]) and np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]), (k, W, H)
            assert rg["status"] == ro["status"] and rg["inliers"] == ro["inliers"], (k, W, H)
    finally:
        for h in hs:
            h.close()
    pr = prs[3]
    s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
    want = None
    for n in range(300):
        with capi.IcpHandle(capi.default_params(pr.intr, iterations=2, max_corr_dist=0.3)) as h:
            if n % 50 == 0:
                T = np.asarray(h.align(s4, t4)["T"])
                want = T if want is None else want
                assert np.array_equal(T, want), n


def test_stamping_ring_can_be_resized_between_runs(gpu_lib):
    """the stamp ring (buffer, modulus, rows per run) is baked into the captured launches as kernel arguments;
    set_stamping(3) -> run -> set_stamping(8) -> run must re-capture instead of replaying a graph that stamps into the freed
    ring with the old modulus.  Eleven runs through the new ring of eight: every row filled, ordered, results unchanged."""
    pr, s4, t4 = _pair(1001, 320, 240)
    iters = 5
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=iters)) as h:
        want = h.align(s4, t4)
        h.set_stamping(3)
        h.set_clouds_host(0, s4, t4)
        for _ in range(2):
            h.run(1); got = h.fetch_results(1)[0]
        assert np.array_equal(got["T_raw"], want["T_raw"])
        h.set_stamping(8)                                            # same handle, stamping stays on, another ring
        for _ in range(11):
            h.run(1); got = h.fetch_results(1)[0]
        ring = h.get_stamps().astype(np.int64)
        assert ring.shape == (8, 2 * iters, 2)
        nn = ring[:, :iters]
        assert (nn[:, :, 0] > 0).all() and (nn[:, :, 1] > nn[:, :, 0]).all()          # every run of the ring has all its NN rows
        assert (ring[1:, 0, 0] > ring[:-1, iters - 1, 1]).all()                       # oldest first
        assert np.array_equal(got["T_raw"], want["T_raw"]) and got["inliers"] == want["inliers"]
        h.set_stamping(2)                                            # shrink: again a fresh capture
        for _ in range(3):
            h.run(1); got = h.fetch_results(1)[0]
        assert h.get_stamps().shape == (2, 2 * iters, 2) and np.array_equal(got["T_raw"], want["T_raw"])


@pytest.mark.parametrize("seed", [1000, 1001])
def test_full_640x480_baseline_md_workload(gpu_lib, seed):
    """BASELINE.md section 4 / SURVEY.md 8(d)'s workload AS SPECIFIED, both parts together: depth noise sigma = 0.0012 z^2
    AND 8x8-pixel Bernoulli holes at p = 0.25 (the two deviations had only been measured separately).  With iid
    noise at that level the reference's own planarity rule (0.01 m, 41 of 49, src/planarFeatures.cpp:118-128) keeps a normal on
    ~1/5 of the targets -- a different problem from the low-noise headline, not a perturbation of it.  640x480, 20
    iterations, as depth images (projective window search) and as clouds: every iterate, every sum, the last indices and
    d2 bit-identical to the kd-tree oracle."""
    pr, s4, t4 = _pair(seed, 640, 480, noise_sigma=0.0012, hole_block=8, hole_prob=0.25)
    ro = O.icp(s4, t4, O.params(pr.intr, iterations=20, nn_method=1))
    assert ro["n_tgt"] < 0.35 * ro["n_src"]                       # the workload's signature: few targets keep a normal
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=20)) as h:
        for depth, trace in ((True, False), (False, False), (True, True)):
            h.set_corr_trace(trace)                               # (with the trace the loop is launched directly, without it as a graph)
            rg = h.align_depth_batch([pr.depth_src], [pr.depth_tgt])[0] if depth else h.align(s4, t4)
            idx, d2 = h.get_correspondences(0)
            Tt, St = h.get_trace(0)
            assert rg["n_src"] == ro["n_src"] and rg["n_tgt"] == ro["n_tgt"]
            assert np.array_equal(idx, ro["idx"]), f"depth={depth}: {(idx != ro['idx']).sum()} index mismatches"
            assert np.array_equal(d2.view(np.uint32), ro["d2"].view(np.uint32))
            assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:20], ro["sums_trace"])
            assert rg["inliers"] == ro["inliers"] and rg["status"] == ro["status"] == 0
            if trace:
                for it in (0, 1, 2, 5, 10, 19):                   # index parity per iteration
                    want, _, _ = O.nn_once(s4, t4, O.params(pr.intr, nn_method=1), T=ro["T_trace"][it], use_normals=True, coarse=(it < 3))    # spec S4c
                    assert np.array_equal(h.get_correspondences_at(it), want), it
    rot_gt, tr_gt = O.pose_error(pr.T_gt, rg["T"])
    assert rot_gt < 1e-2 and tr_gt < 3e-2, (rot_gt, tr_gt)       # noise floor of THIS workload: six times the headline's sigma and only the near
                                                                 # fifth of the target keeps a normal (oracle: 7 mrad / 2.2 cm on seed 1000, 3 mrad / 1 cm on 1001)


def test_config5_dense_1280x960_eight_emulated_ranks(gpu_lib):
    """BASELINE config 5's 8-GPU leg as far as ONE device allows: the 1280x960 pair of seed 2000, 20
    iterations, source rows sharded over EIGHT ranks = eight handles on this GPU, each searching its row band against the
    whole target; the per-iteration exchange is the integer sum of the ranks' 36-word partials (what ncclAllReduce(SUM) of
    int64 computes; here formed on the host).  Every rank ends with the same pose, bit-identical to the UNSHARDED run on
    the GPU and to the oracle; the ranks' index bands stitched together are the unsharded indices."""
    from slam3d_gx_amd import shard
    world, iters = 8, 20
    pr, s4, t4 = _pair(2000, 1280, 960)
    ro = O.icp(s4, t4, O.params(pr.intr, iterations=iters, nn_method=1))
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=iters)) as h1:
        whole = h1.align(s4, t4)
        idx_whole = h1.get_correspondences(0)[0]
    hs = [capi.IcpHandle(capi.default_params(pr.intr, iterations=iters)) for _ in range(world)]
    try:
        bands = [shard.dense_row_range(pr.intr.height, world, r) for r in range(world)]
        assert bands[0][0] == 0 and bands[-1][1] == pr.intr.height and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
        for r, h in enumerate(hs):
            h.set_clouds_host(0, s4, t4)
            h.dense_set_rows(*bands[r])
            h.dense_begin(None)
        total = None
        for _ in range(iters):
            parts = [h.dense_partial() for h in hs]
            total = np.sum(np.stack(parts), axis=0, dtype=np.int64)
            for h in hs:
                h.dense_update(total)
        res = [h.dense_finish(total) for h in hs]
        idx = [h.get_correspondences(0)[0] for h in hs]
        n_src = [r["n_src"] for r in res]
    finally:
        for h in hs:
            h.close()
    for r in res:
        assert np.array_equal(r["T_raw"], res[0]["T_raw"]) and r["inliers"] == res[0]["inliers"]
    assert np.array_equal(res[0]["T_raw"], whole["T_raw"]) and res[0]["inliers"] == whole["inliers"]       # == the unsharded GPU run
    assert np.array_equal(res[0]["T_raw"], ro["T_trace"][-1]) and res[0]["inliers"] == ro["inliers"]       # == the oracle
    assert sum(n_src) == whole["n_src"] == ro["n_src"]
    W = pr.intr.width
    merged = np.full_like(idx_whole, -1)
    for (r0, r1), ix in zip(bands, idx):
        merged[r0 * W:r1 * W] = ix[r0 * W:r1 * W]
        outside = np.ones(ix.size, dtype=bool); outside[r0 * W:r1 * W] = False
        assert (ix[outside] == -1).all()                          # a rank reports nothing outside its band
    assert np.array_equal(merged, idx_whole) and np.array_equal(merged, ro["idx"])


def test_config4_shape_eight_ranks_of_64_pairs_on_one_device(gpu_lib, tmp_path):
    """BASELINE config 4's shape: 512 frame pairs of 640x480 (seeds 1000..1511) sharded over EIGHT ranks,
    64 per rank in one launch sequence, the step's 512 pose records gathered on every rank -- eight processes on this one
    GPU over gloo (RCCL needs a GPU per rank; everything else is the code the driver's 8-GPU run executes), started the way
    the driver starts it: a plain `python bench.py --gpus 8 ...`.  The gathered table must be in pair order (rank r holds
    the contiguous block of seeds 1000 + 64 r ...) and sampled pairs bit-identical to the oracle."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    dump = str(tmp_path / "table.json")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--pairs", "64",
           "--pairs-per-step", "64", "--pool", "64", "--in-flight", "1", "--no-cpu-baseline", "--no-bruteforce", "--profile-aligns", "1",
           "--overlap-aligns", "0", "--dist-backend", "gloo", "--one-device", "--dump-table", dump, "--legs-file", str(tmp_path / "legs.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=2400, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    last = out.stdout.rstrip("\n").splitlines()[-1]
    assert len(last.encode()) < 6000
    d = json.loads(last)
    d["per_rank"] = json.load(open(tmp_path / "legs.json"))["per_rank"]
    assert d["n_gpus"] == 8 and d["value"] > 0 and d["scaling"] == "weak" and d["survey_8d"]["gpus"] == 8
    assert d["config"]["gathered_pose_records"] == 512 and d["config"]["pairs_per_launch"] == 64
    assert d["config"]["coarse_iterations"] == 3 and d["config"]["synthetic_workload"].startswith("BASELINE.md section 4")
    assert [r["rank"] for r in d["per_rank"]] == list(range(8)) and all(r["h2d_GBps"] > 0 and r["enqueue_us_per_alignment"] > 0 and r["gather_us_per_step"] > 0 for r in d["per_rank"])
    assert d["config"]["gathered_seeds"] == {"first": [1000, 1001], "last": [1510, 1511], "ascending": True}
    tab = json.load(open(dump))
    assert tab["seeds"] == list(range(1000, 1512)) and tab["world"] == 8 and tab["pairs_per_rank"] == 64
    assert all(s == 0 for s in tab["status"]), [i for i, s in enumerate(tab["status"]) if s][:8]
    for k in (0, 63, 64, 200, 383, 511):                          # both ends of rank blocks and interior pairs
        pr, s4, t4 = _pair(1000 + k, 640, 480, noise_sigma=0.0012, hole_block=8, hole_prob=0.25)      # bench.py's default workload: BASELINE.md section 4's
        ro = O.icp(s4, t4, O.params(pr.intr, iterations=20, nn_method=1))
        assert np.array_equal(np.array(tab["T"][k]).reshape(4, 4), ro["T_trace"][-1]), k
        assert tab["inliers"][k] == ro["inliers"], k


@pytest.mark.parametrize("coarse,iters", [(0, 6), (1, 6), (5, 6), (3, 1), (3, 2), (9, 4)])
@pytest.mark.parametrize("batch", [1, 9])
def test_coarse_iteration_counts_and_short_runs(gpu_lib, coarse, iters, batch):
    """Spec S4c for other settings than the default three: no coarse iteration at all, more coarse iterations than the run
    has (every iteration but the last is coarse then), runs of one and two iterations -- on the cooperative build (one pair
    per launch) and the throughput build (nine), with depth inputs (window search) and an initial guess for one pair: every
    iterate, the sums and the last indices equal the oracle's."""
    prs = [synth.make_pair(4100 + i, 200, 150, holes=True) for i in range(batch)]
    intr = prs[0].intr
    Ti = np.stack([synth.pose_from_seed(9 + i, 1.5, 0.03) if i % 2 else np.eye(4) for i in range(batch)])
    kw = dict(iterations=iters, coarse_iterations=coarse)
    with capi.IcpHandle(capi.default_params(intr, max_batch=batch, **kw)) as h:
        res = h.align_depth_batch([p.depth_src for p in prs], [p.depth_tgt for p in prs], Ti)
        got = [(h.get_trace(b), h.get_correspondences(b)[0]) for b in range(batch)]
    for b in (0, batch - 1):
        po = O.params(intr, nn_method=0, **kw)
        s4, t4 = O.backproject(prs[b].depth_src, po), O.backproject(prs[b].depth_tgt, po)
        ro = O.icp(s4, t4, po, T_init=Ti[b])
        (Tt, St), idx = got[b]
        assert np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:iters], ro["sums_trace"]), (coarse, iters, b)
        assert np.array_equal(idx, ro["idx"]) and res[b]["inliers"] == ro["inliers"] and res[b]["status"] == ro["status"]
    if iters >= 2 and coarse >= 1:      # the first iteration really was a coarse one: about a quarter of the rows
        assert St[0][27] < 0.45 * St[-1][27]


@pytest.mark.parametrize("window,min_in", [(3, 7), (5, 20), (9, 60)])
def test_other_normal_windows_than_7x7(gpu_lib, window, min_in):
    """Spec S2 with the window radius taken at run time (k_normals<0>: the 7x7 default has its own unrolled instance): 3x3, 5x5
    and 9x9 windows (9 = the largest the staging supports), frames with holes and an odd size -- normals and planar flags
    bit-identical to the oracle, and a short alignment on top of them too."""
    pr, s4, t4 = _pair(5200 + window, 200, 150, holes=True)
    kw = dict(normal_window=window, normal_min_inliers=min_in, iterations=5)
    po = O.params(pr.intr, nn_method=0, **kw)
    n_or = O.normals(t4, po)
    ro = O.icp(s4, t4, po)
    with capi.IcpHandle(capi.default_params(pr.intr, **kw)) as h:
        r = h.align(s4, t4)
        _, _, n_gpu = h.get_clouds(0, normals=True)
        idx, _ = h.get_correspondences(0)
        Tt, St = h.get_trace(0)
    assert (n_or[..., 3] > 0.5).sum() > 2000
    assert np.array_equal(n_gpu.view(np.uint32), n_or.view(np.uint32))
    assert np.array_equal(idx, ro["idx"]) and np.array_equal(Tt.reshape(-1, 4, 4), ro["T_trace"]) and np.array_equal(St[:5], ro["sums_trace"])
    assert r["inliers"] == ro["inliers"]

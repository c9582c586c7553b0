"""f-2: batched RANSAC plane segmentation -- oracle properties, golden vectors, HIP vs golden.

The reference's step is pcl::SACSegmentation inside GraphicEnd::extractPlanesAndGenerateImage
(src/GraphicEnd.cpp:353-430); PCL is not in the tree and draws from rand(), so parity is UNPINNED: the
goldens come from oracle/seg_oracle.c (tests/golden/make_golden.py segmentation).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from slam3d_gx_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
G = json.load(open(os.path.join(HERE, "golden", "segmentation_golden.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cloud(c):
    pr = synth.make_pair(c["seed"], c["width"], c["height"])
    assert pr.sha256() == c["depth_sha256"]
    return pr, synth.backproject_numpy(pr.depth_src, pr.intr)


def _check(c, planes, labels):
    assert len(planes) == c["nplanes"]
    assert sha(labels) == c["labels_sha256"]
    assert int((labels == -1).sum()) == c["unassigned"] and int((labels == -2).sum()) == c["invalid"]
    for p, g in zip(planes, c["planes"]):
        assert p["count"] == g["count"]
        assert [float(x).hex() for x in p["coeff"]] == g["coeff"]
        assert [float(x).hex() for x in p["centroid"]] == g["centroid"]


@pytest.mark.parametrize("c", G["cases"], ids=lambda c: f"{c['width']}x{c['height']}-{c['seed']}")
def test_oracle_reproduces_segmentation_golden(c):
    _, s4 = _cloud(c)
    _check(c, *O.segment_planes(s4, seed=c["seed"]))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference fixtures not present")
def test_oracle_segmentation_on_reference_frame():
    from PIL import Image
    d = np.array(Image.open(os.path.join(REF, "data/exp1/dep/1.png"))).astype(np.uint16)
    c1 = O.backproject(d, O.params(synth.Intrinsics()))
    planes, labels = O.segment_planes(c1, seed=1)
    _check(G["reference_dep1"], planes, labels)
    assert planes[0]["count"] > 100000 and abs(planes[0]["coeff"][1]) > 0.99      # the dominant horizontal surface


def test_labels_are_consistent_with_planes():
    pr = synth.make_pair(1001, 320, 240)
    s4 = synth.backproject_numpy(pr.depth_src, pr.intr).reshape(-1, 4)
    planes, labels = O.segment_planes(s4, seed=5)
    valid = np.isfinite(s4[:, 2]) & (s4[:, 2] > 0) & (s4[:, 2] <= 7.0)
    assert np.array_equal(labels == -2, ~valid)
    assert 2 <= len(planes) <= 3
    rest = valid.copy()
    for r, p in enumerate(planes):
        a = p["coeff"].astype(np.float64)
        assert a[3] >= 0 and abs(np.linalg.norm(a[:3]) - 1) < 1e-6              # src/GraphicEnd.cpp:383-387
        e = np.abs(s4[:, :3].astype(np.float64) @ a[:3] + a[3])
        m = labels == r
        assert m.sum() == p["count"] and (e[m] <= 0.08 + 1e-6).all()
        # every still-unassigned point within the threshold was taken by this plane (inliers are removed, :419-420)
        assert not (rest & ~m & (e < 0.08 - 1e-6)).any()
        rest &= ~m
    assert np.array_equal(labels == -1, rest)
    # loop rule (:372): it stopped because <= 20 % is left or max_planes was reached
    assert len(planes) == 3 or rest.sum() <= 0.2 * valid.sum()


def test_analytic_planes_are_recovered():
    """Noise-free points on three known planes: coefficients within 1e-4 whatever hypotheses were drawn."""
    # (KAT of SURVEY.md 8c(1) style: analytic ground truth, not an oracle-vs-oracle comparison)
    rng = np.random.default_rng(3)
    W, H = 160, 120
    N = W * H
    c = np.full((N, 4), np.nan, dtype=np.float32)
    n0, n1 = N // 2, N // 2 + N // 4
    c[:n0, 0] = rng.uniform(-2, 2, n0); c[:n0, 1] = rng.uniform(-1, 1, n0); c[:n0, 2] = 4.5             # z = 4.5
    c[n0:n1, 0] = -2.0; c[n0:n1, 1] = rng.uniform(-1, 1, n1 - n0); c[n0:n1, 2] = rng.uniform(1, 4, n1 - n0)  # x = -2
    k = N - n1 - 500
    c[n1:n1 + k, 0] = rng.uniform(-2, 2, k); c[n1:n1 + k, 1] = 1.2; c[n1:n1 + k, 2] = rng.uniform(1, 4, k)   # y = 1.2
    c[:, 3] = 1.0
    # a tight threshold keeps the points of the other planes near the intersection lines out of the fits
    planes, labels = O.segment_planes(c, seed=9, plane_percent=0.01, distance_threshold=0.002)
    exp = [np.array([0, 0, -1, 4.5]), np.array([1, 0, 0, 2.0]), np.array([0, -1, 0, 1.2])]
    assert len(planes) == 3
    for p, e in zip(planes, exp):
        assert np.abs(p["coeff"] - e).max() < 1e-4
    assert planes[0]["count"] >= n0 and (labels[-500:] == -2).all()


def test_loop_parameters_and_determinism():
    pr = synth.make_pair(1000, 160, 120)
    s4 = synth.backproject_numpy(pr.depth_src, pr.intr)
    a = O.segment_planes(s4, seed=2)
    b = O.segment_planes(s4, seed=2)
    assert np.array_equal(a[1], b[1]) and len(a[0]) == len(b[0])
    one = O.segment_planes(s4, seed=2, max_planes=1)
    assert len(one[0]) == 1 and np.array_equal(one[0][0]["coeff"], a[0][0]["coeff"])
    # plane_percent = 0.9: the dominant plane alone leaves less than 90 % -> one plane only
    stop = O.segment_planes(s4, seed=2, plane_percent=0.9)
    assert len(stop[0]) == 1
    # another seed finds the same dominant plane (not the same bits)
    other = O.segment_planes(s4, seed=77)
    assert np.abs(other[0][0]["coeff"] - a[0][0]["coeff"]).max() < 2e-2
    # empty cloud
    e = O.segment_planes(np.full_like(s4, np.nan), seed=2)
    assert e[0] == [] and (e[1] == -2).all()


@pytest.mark.gpu
@pytest.mark.parametrize("persist", ["0", "1"])
@pytest.mark.parametrize("c", G["cases"], ids=lambda c: f"{c['width']}x{c['height']}-{c['seed']}")
def test_hip_segmentation_reproduces_golden(gpu_lib, c, persist, monkeypatch):
    """Both forms of the single-frame pass: the launches (default) and the ONE persistent launch of round 6 (SLAM3D_SEG_PERSIST=1:
    pixels resident in LDS, labels of drawn pixels decided from the planes, grid barriers; measured slower and kept off, plane_seg.hpp)
    -- the same planes and labels as the golden vectors."""
    from slam3d_gx_amd import capi
    monkeypatch.setenv("SLAM3D_SEG_PERSIST", persist)
    pr, s4 = _cloud(c)
    with capi.IcpHandle(capi.default_params(pr.intr, max_batch=1)) as h:
        planes, labels = h.segment_planes(s4, h.seg_params(seed=c["seed"]))
    _check(c, planes, labels)


@pytest.mark.gpu
def test_hip_segmentation_batched_path_equals_the_single_frame_path(gpu_lib):
    """A frame alone takes the fused three-launch rounds, a batch the five-launch ones (csrc/plane_seg.hpp, round 4): both run on
    the same double-buffered round state and must give every frame the same planes and labels -- five different frames in one
    call against each of them alone (which the golden tests pin to the oracle)."""
    import torch
    from slam3d_gx_amd import capi, synth
    frames = []
    for seed in (3001, 3002, 3003, 3004, 3005):
        pr = synth.make_pair(seed, 320, 240)
        frames.append(synth.backproject_numpy(pr.depth_src, pr.intr))
    intr = synth.make_pair(3001, 320, 240).intr
    N = 320 * 240
    with capi.IcpHandle(capi.default_params(intr, max_batch=8)) as h:
        sp = h.seg_params(seed=11)
        alone = [h.segment_planes(f, sp) for f in frames]
        d = torch.from_numpy(np.stack([f.reshape(N, 4) for f in frames])).to("cuda:0")
        d_lab = torch.zeros((len(frames), N), dtype=torch.int32, device="cuda:0")
        ptrs = [d[i].data_ptr() for i in range(len(frames))]
        batch = h.segment_planes_device(ptrs, sp, d_lab.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        lab = d_lab.cpu().numpy()
    assert len(batch) == len(frames)
    for i, (planes1, labels1) in enumerate(alone):
        assert len(batch[i]) == len(planes1) and len(planes1) >= 1
        for a, b in zip(batch[i], planes1):
            assert np.array_equal(a["coeff"], b["coeff"]) and a["count"] == b["count"] and np.array_equal(a["centroid"], b["centroid"])
        assert np.array_equal(lab[i], np.asarray(labels1).reshape(-1))

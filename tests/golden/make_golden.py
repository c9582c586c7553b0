#!/usr/bin/env python3
"""Generates tests/golden/*.json from the CPU oracle (run in the authoring container).

  synthetic_golden.json   oracle outputs for deterministic synthetic pairs (inputs are regenerated
                          by slam3d_gx_amd.synth anywhere; their SHA-256 is stored)
  reference_golden.json   DERIVED outputs only (hashes, poses, counts) of the oracle run on the
                          reference's real fixtures data/exp1/dep/{1,2}.png and bin/dep_1.png, read from
                          /root/reference.  No reference data or source is copied (GPLv3).
  segmentation_golden.json  f-2 plane segmentation (oracle/seg_oracle.c) on synthetic frames + derived outputs on
                          the reference's data/exp1/dep/1.png
Floats are stored as C99 hex strings (bit-exact round trip).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from slam3d_gx_amd import synth  # noqa: E402

REF = "/root/reference"


def hx(a):
    return [float(x).hex() for x in np.asarray(a, dtype=np.float64).reshape(-1)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def summarize(r, p):
    valid = np.nonzero(r["idx"] >= 0)[0]
    return dict(
        n_src=r["n_src"], n_tgt=r["n_tgt"], inliers=r["inliers"], status=r["status"], norm=float(r["norm"]).hex(),
        rmse=float(r["rmse"]).hex(), T_final=hx(r["T_trace"][-1]), T_iter1=hx(r["T_trace"][1]),
        sums_first=hx(r["sums_trace"][0]), sums_last=hx(r["sums_trace"][-1]),
        idx_sha256=sha(r["idx"]), d2_sha256=sha(r["d2"]), n_corr=int(valid.size),
        idx_first64=[[int(i), int(r["idx"][i])] for i in valid[:64]],
        idx_last64=[[int(i), int(r["idx"][i])] for i in valid[-64:]],
        iterations=p.iterations, estimator=p.estimator)


def synthetic():
    cases = []
    for (w, h, seeds, iters, method) in ((160, 120, (1000, 1001, 1002, 1003), 5, 0), (640, 480, (1000,), 20, 1)):
        for seed in seeds:
            pr = synth.make_pair(seed, w, h)
            s4 = synth.backproject_numpy(pr.depth_src, pr.intr)
            t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
            for est in (0, 1):
                p = O.params(pr.intr, estimator=est, iterations=iters, nn_method=method)
                r = O.icp(s4, t4, p)
                c = dict(seed=seed, width=w, height=h, depth_sha256=pr.sha256(), T_gt=hx(pr.T_gt))
                c.update(summarize(r, p))
                if est == 0:
                    c["normals_sha256"] = sha(O.normals(t4, p))
                cases.append(c)
                print("synthetic", w, h, seed, est, c["inliers"], c["status"])
    json.dump(dict(generator="tests/golden/make_golden.py", cases=cases), open(os.path.join(HERE, "synthetic_golden.json"), "w"), indent=1)


def read_png16(path):
    from PIL import Image
    a = np.array(Image.open(path))
    assert a.dtype in (np.uint16, np.int32), a.dtype
    return a.astype(np.uint16)


def reference():
    if not os.path.isdir(REF):
        print("no /root/reference: skipping reference_golden.json")
        return
    intr = synth.Intrinsics()    # src/convert2PCD.cpp:19-23 -- the intrinsics the fixtures were made with
    d1 = read_png16(os.path.join(REF, "data/exp1/dep/1.png"))
    d2 = read_png16(os.path.join(REF, "data/exp1/dep/2.png"))
    db = read_png16(os.path.join(REF, "bin/dep_1.png"))
    out = dict(generator="tests/golden/make_golden.py", note="derived outputs only; inputs are read from /root/reference",
               inputs={k: dict(sha256_prefix=hashlib.sha256(open(os.path.join(REF, f), "rb").read()).hexdigest()[:16],
                               nonzero=int((d > 0).sum()))
                       for k, f, d in (("dep1", "data/exp1/dep/1.png", d1), ("dep2", "data/exp1/dep/2.png", d2), ("bin_dep_1", "bin/dep_1.png", db))})
    p = O.params(intr)
    c1, c2, cb = (O.backproject(d, p) for d in (d1, d2, db))
    out["backproject_sha256"] = dict(dep1=sha(c1), dep2=sha(c2), bin_dep_1=sha(cb))
    # config 1 plumbing: organized normals / planarity on bin/dep_1.png (row a7)
    nb = O.normals(cb, p)
    out["config1_bin_dep_1"] = dict(valid_points=int(np.isfinite(cb[..., 2]).sum()), planar_pixels=int((nb[..., 3] > 0).sum()),
                                    normals_sha256=sha(nb))
    # wide-baseline real pair: GPU == oracle equality test only (SURVEY.md App. D caveat)
    pairs = {}
    for est in (0, 1):
        pp = O.params(intr, estimator=est, iterations=10, nn_method=1)
        r = O.icp(c1, c2, pp)
        pairs[str(est)] = summarize(r, pp)
        print("real pair est", est, r["inliers"], r["status"], r["norm"])
    out["real_pair_dep1_to_dep2"] = pairs
    json.dump(out, open(os.path.join(HERE, "reference_golden.json"), "w"), indent=1)


def seg_summary(planes, labels):
    return dict(nplanes=len(planes), labels_sha256=sha(labels), unassigned=int((labels == -1).sum()),
                invalid=int((labels == -2).sum()),
                planes=[dict(coeff=[float(x).hex() for x in p["coeff"]], centroid=[float(x).hex() for x in p["centroid"]],
                             count=p["count"]) for p in planes])


def segmentation():
    """f-2 plane segmentation: oracle/seg_oracle.c on synthetic frames and (derived outputs only) on the
    reference's data/exp1/dep/1.png."""
    out = dict(generator="tests/golden/make_golden.py segmentation", cases=[])
    for (w, h, seed) in ((160, 120, 1000), (320, 240, 1001), (640, 480, 1000), (640, 480, 1002)):
        pr = synth.make_pair(seed, w, h)
        s4 = synth.backproject_numpy(pr.depth_src, pr.intr)
        planes, labels = O.segment_planes(s4, seed=seed)
        c = dict(seed=seed, width=w, height=h, depth_sha256=pr.sha256())
        c.update(seg_summary(planes, labels))
        out["cases"].append(c)
        print("seg", w, h, seed, [p["count"] for p in planes])
    if os.path.isdir(REF):
        d1 = read_png16(os.path.join(REF, "data/exp1/dep/1.png"))
        c1 = O.backproject(d1, O.params(synth.Intrinsics()))
        planes, labels = O.segment_planes(c1, seed=1)
        out["reference_dep1"] = seg_summary(planes, labels)
        print("seg reference dep1", [(p["count"], p["coeff"]) for p in planes])
    json.dump(out, open(os.path.join(HERE, "segmentation_golden.json"), "w"), indent=1)


def read_pcd16(path):
    """binary PCD v0.7 with FIELDS x y z rgba -> (n,4) float32 records (rgba bits in column 3)"""
    raw = open(path, "rb").read()
    k = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
    head = raw[:k].decode("ascii", "replace")
    n = int([ln for ln in head.splitlines() if ln.startswith("POINTS")][0].split()[1])
    assert "FIELDS x y z rgba" in head
    return np.frombuffer(raw, dtype=np.float32, count=4 * n, offset=k).reshape(n, 4).copy()


def voxel():
    """f-1 PassThrough + VoxelGrid(0.03): synthetic clouds with seeded colours + (derived outputs only) the
    reference's data/exp1/pcd/{1,2}.pcd at its own operating point (grid_leaf 0.03, z_filter 7.0)."""
    out = dict(generator="tests/golden/make_golden.py voxel", cases=[])
    for (w, h, seed) in ((160, 120, 1000), (640, 480, 1001)):
        pr = synth.make_pair(seed, w, h)
        c = synth.backproject_numpy(pr.depth_src, pr.intr).reshape(-1, 4).copy()
        c[:, 3] = np.random.default_rng(seed).integers(0, 2 ** 32, c.shape[0], dtype=np.uint64).astype(np.uint32).view(np.float32)
        v = O.voxel_grid(c, 0.03, 7.0)
        out["cases"].append(dict(seed=seed, width=w, height=h, depth_sha256=pr.sha256(), leaf=0.03, voxels=int(v.shape[0]),
                                 out_sha256=sha(v), first=[float(x).hex() for x in v[0, :3]], last=[float(x).hex() for x in v[-1, :3]]))
        print("voxel", w, h, seed, v.shape[0])
    if os.path.isdir(REF):
        for k in (1, 2):
            p = read_pcd16(os.path.join(REF, f"data/exp1/pcd/{k}.pcd"))
            v = O.voxel_grid(p, 0.03, 7.0)
            out[f"reference_pcd{k}"] = dict(points=int(p.shape[0]), voxels=int(v.shape[0]), out_sha256=sha(v),
                                            passthrough_kept=int((p[:, 2] <= 7.0).sum()))
            print("voxel reference pcd", k, p.shape[0], "->", v.shape[0])
    json.dump(out, open(os.path.join(HERE, "voxel_golden.json"), "w"), indent=1)


if __name__ == "__main__":
    which = sys.argv[1:] or ["synthetic", "reference", "segmentation", "voxel"]
    for w in which:
        {"synthetic": synthetic, "reference": reference, "segmentation": segmentation, "voxel": voxel}[w]()

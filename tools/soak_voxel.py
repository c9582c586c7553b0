#!/usr/bin/env python3
"""Randomised parity soak of the voxel grid (developer tool, needs an MI355X): organized frames and point lists of random sizes, random leaves
(dense and general ordering paths), clouds shifted / scaled out of the dense key range, NaN holes, duplicated points, single calls and batches
with mixed frames on one handle -- every record must be the oracle's bits.  usage: tools/soak_voxel.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from slam3d_gx_amd import capi, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
W, H = 320, 240
intr = synth.Intrinsics(width=W, height=H)
base = []
for seed in range(40, 46):
    pr = synth.make_pair(seed, W, H)
    c = synth.backproject_numpy(pr.depth_src, pr.intr).reshape(-1, 4).copy()
    c[:, 3] = rng.integers(0, 2 ** 32, c.shape[0], dtype=np.uint64).astype(np.uint32).view(np.float32)
    base.append(c)


def make_cloud():
    c = base[int(rng.integers(0, len(base)))].copy()
    kind = rng.random()
    if kind < 0.35:
        pass                                                      # organized frame as it is
    elif kind < 0.6:
        c = c[np.isfinite(c[:, 2])]                               # the PCD form: a list without the invalid pixels
        c = c[: int(rng.integers(1, len(c) + 1))]
    elif kind < 0.75:
        c = c[rng.permutation(len(c))[: int(rng.integers(1, len(c) + 1))]]       # shuffled list
    elif kind < 0.85:
        c = np.repeat(c[rng.integers(0, len(c), 50)], int(rng.integers(1, 400)), axis=0)      # heavy duplicates: few voxels, many points
    else:
        c[rng.random(len(c)) < 0.5] = np.nan                      # organized with holes
    if rng.random() < 0.25:
        c[:, :3] *= np.float32(rng.choice([0.1, 3.0, 40.0]))      # tiny / huge scenes (z filter drops most of the huge ones)
    if rng.random() < 0.25:
        c[:, 0] += np.float32(rng.choice([-30.0, 7.0, 12.0, 1000.0])); c[:, 1] += np.float32(rng.choice([0.0, -5.0, 9.0]))
    return np.ascontiguousarray(c[: W * H], dtype=np.float32)


bad = 0
t0 = time.time()
dense = general = 0
with capi.IcpHandle(capi.default_params(intr, max_batch=1)) as h:
    st = torch.cuda.Stream()
    for case in range(n_cases):
        leaf = float(rng.choice([0.002, 0.007, 0.02, 0.03, 0.03, 0.05, 0.3, 5.0]))
        B = int(rng.choice([1, 1, 2, 3, 5, 9, 17]))
        clouds = [make_cloud() for _ in range(B)]
        if B > 1 and rng.random() < 0.5:                          # a batch of organized frames takes the tile insert: all must be full
            clouds = [base[int(rng.integers(0, len(base)))].copy() for _ in range(B)]
            for c in clouds:
                if rng.random() < 0.3: c[:, 0] += np.float32(20.0)
        wants = [O.voxel_grid(c, leaf, 7.0) for c in clouds]
        ds = [torch.from_numpy(c).to("cuda:0") for c in clouds]
        outs = [torch.zeros((max(1, len(c)), 4), dtype=torch.float32, device="cuda:0") for c in clouds]
        stream = st.cuda_stream if rng.random() < 0.5 else torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        if B == 1 and rng.random() < 0.5:
            got = [h.voxel_grid(clouds[0], leaf=leaf)]
        else:
            ms = h.voxel_grid_batch_device([d.data_ptr() for d in ds], [len(c) for c in clouds], [o.data_ptr() for o in outs], leaf, stream)
            torch.cuda.synchronize()
            got = [o[:m].cpu().numpy() for o, m in zip(outs, ms)]
        for k, (g, w) in enumerate(zip(got, wants)):
            if g.shape != w.shape or not np.array_equal(g.view(np.uint32), w.view(np.uint32)):
                bad += 1
                print("MISMATCH", dict(case=case, leaf=leaf, B=B, k=k, n=len(clouds[k]), got=g.shape, want=w.shape), flush=True)
    dense, general = h.voxel_grid_path_counts()
print(f"{n_cases} voxel cases, {bad} mismatches, dense-only calls {dense}, calls with the general path {general}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Developer tool: per-wave statistics of the tile-pruned NN kernel (needs SLAM3D_NN_DEBUG=1): phase durations, work
counters, and WHERE / WHEN every wave ran (XCD, CU, start and end relative to the launch) -- the launch lasts as long
as its last wave, so the end-time distribution per XCD / CU is what explains the kernel time."""
import os, sys
os.environ["SLAM3D_NN_DEBUG"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam3d_gx_amd import capi, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
if seed < 0:      # the reference's Kinect pair (tests/golden/kinect): dep1 -> dep2
    from PIL import Image
    kin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kinect")
    pr = synth.FramePair(-1, synth.Intrinsics(), np.array(Image.open(os.path.join(kin, "exp1_dep_1.png"))).astype(np.uint16),
                         np.array(Image.open(os.path.join(kin, "exp1_dep_2.png" if seed == -1 else "exp1_dep_1.png"))).astype(np.uint16), np.eye(4))
else:
    pr = synth.make_pair(seed, int(os.environ.get("NNDBG_W", "640")), int(os.environ.get("NNDBG_H", "480")), **(dict(noise_sigma=0.0012, hole_block=8, hole_prob=0.25) if os.environ.get("NNDBG_BMD") else {}))
s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
for it in (1, iters):
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=it, estimator=int(os.environ.get("NNDBG_EST", "0")), plane_flags=int(os.environ.get("NNDBG_FLAGS", "0")))) as h:
        h.align_depth_batch([pr.depth_src], [pr.depth_tgt])      # depth inputs: the projective window search takes part
        d = h.get_nn_debug()
    act = d[:, 4] > 0
    d = d[act].copy()
    rt0 = d[:, 10].min()
    rs, re_ = (d[:, 10] - rt0) * 10, (d[:, 11] - rt0) * 10          # ns since the first wave started (100 MHz counter)
    t0 = 0
    print(f"--- last of {it} iteration(s): {act.sum()} active waves; launch span (first start -> last end) {re_.max()} ns")
    dc = d[:, 9] >> 32
    nc_all, nv_all = ((d[:, 18] >> 32) & 0xff).copy(), ((dc >> 24) & 0xff).copy()
    why = [int((dc & 0xff).sum()), int(((dc >> 8) & 0xff).sum()), int(((dc >> 16) & 0xff).sum())]
    d[:, 18] &= 0xffffffff
    d1 = d[d[:, 1] > 0]
    lo32 = lambda x: x & 0xffffffff
    hi32 = lambda x: x >> 32
    for name, v in (("lifetime", d1[:, 4] - d1[:, 0]), ("start ns", rs[d[:, 1] > 0]), ("end ns", re_[d[:, 1] > 0]), ("prologue", d1[:, 1] - d1[:, 0]), ("own 5 tiles", d1[:, 2] - d1[:, 1]),
                    ("publish+items (to barrier 2)", d1[:, 3] - d1[:, 2]),
                    ("  publish", d1[:, 12] - d1[:, 2]), ("  wait barrier 1", d1[:, 13] - d1[:, 12]), ("  phase A (cells)", d1[:, 14] - d1[:, 13]),
                    ("  wait barrier A", d1[:, 16] - d1[:, 14]), ("  phase B (tiles)", d1[:, 17] - d1[:, 16]), ("  wait barrier 2", d1[:, 3] - d1[:, 17]),
                    ("  cells done by wave", lo32(d1[:, 15])), ("  cell items of block", hi32(d1[:, 15])), ("  tile items of block", d1[:, 18]),
                    ("epilogue", d1[:, 4] - d1[:, 3]), ("tiles scanned", lo32(d1[:, 5])), ("candidates", lo32(d1[:, 6])), ("batches", lo32(d1[:, 7])),
                    ("cells swept", hi32(d1[:, 5])), ("fine hits", hi32(d1[:, 6])), ("refined hits", hi32(d1[:, 7]))):
        print(f"{name:14s} mean {v.mean():10.1f}  p50 {np.percentile(v,50):9.0f}  p90 {np.percentile(v,90):9.0f}  p99 {np.percentile(v,99):9.0f}  max {v.max():9.0f}")
    nc, nv = nc_all, nv_all
    if nv.sum():
        full = (nc == nv) & (nv > 0)
        print(f"certified lanes {nc.sum()} of {nv.sum()} valid ({100.0 * nc.sum() / nv.sum():.1f} %); waves with every valid lane certified: {full.sum()} of {(nv > 0).sum()} "
              f"({100.0 * full.sum() / max(1, (nv > 0).sum()):.1f} %); uncertified lanes per not-fully-certified wave: mean {(nv - nc)[~full & (nv > 0)].mean() if (~full & (nv > 0)).any() else 0:.1f}; "
              f"why not: no clearance {why[0]}, match left the gate {why[1]}, clearance used up {why[2]}")
    hb = d[:, 19]
    nit = (hb & 0xff).sum()
    if nit:
        print(f"phase B items processed {nit}: active lanes per item <=2: {((hb >> 8) & 0xff).sum() / nit:.2f}  <=4: {((hb >> 16) & 0xff).sum() / nit:.2f}  "
              f"<=8: {((hb >> 24) & 0xff).sum() / nit:.2f}  <=16: {((hb >> 32) & 0xff).sum() / nit:.2f}  mean {(hb >> 40).sum() / nit:.1f}")
    hw = d[:, 8] & 0xffffffff
    xcc = (d[:, 8] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    end = re_
    print("per XCD: waves, mean end, max end, sum lifetime")
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"  xcd {x}: {m.sum():5d} waves  end mean {end[m].mean():9.0f} max {end[m].max():9.0f}  work {(d[m, 4] - d[m, 0]).sum():12d}")
    ids, inv = np.unique(cuid, return_inverse=True)
    cu_end = np.array([end[inv == k].max() for k in range(len(ids))]); cu_n = np.array([(inv == k).sum() for k in range(len(ids))])
    cu_work = np.array([(d[inv == k, 4] - d[inv == k, 0]).sum() for k in range(len(ids))])
    print(f"CUs seen {len(ids)}; waves per CU min {cu_n.min()} mean {cu_n.mean():.1f} max {cu_n.max()}; CU end time: min {cu_end.min()} p10 {np.percentile(cu_end,10):.0f} "
          f"p50 {np.percentile(cu_end,50):.0f} p90 {np.percentile(cu_end,90):.0f} max {cu_end.max()}; CU work (sum of wave lifetimes) min {cu_work.min()} p50 {np.percentile(cu_work,50):.0f} max {cu_work.max()}")
    print(f"corr(CU end, waves per CU) = {np.corrcoef(cu_end, cu_n)[0,1]:.2f}   corr(CU end, CU work) = {np.corrcoef(cu_end, cu_work)[0,1]:.2f}")
    late = np.argsort(end)[-12:]
    print("latest waves: (end, lifetime, items-phase, fine hits, candidates, xcd, cu)")
    for i in late:
        print(f"   {end[i]:8d} {d[i,4]-d[i,0]:8d} {d[i,3]-d[i,2]:8d} {hi32(d[i,6]):4d} {lo32(d[i,6]):5d}  xcd {xcc[i]} cu {cuid[i]}")

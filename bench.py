#!/usr/bin/env python3
"""bench.py -- ICP iterations/s of the MI355X plane-ICP path (BASELINE.json metric).

A *step* is one pass of the hot path over one batch of synthetic frame pairs that are already
resident in HBM: preprocessing (normals, tiles) + `iterations` ICP iterations + the pose records
back on the host.  Workload at N=1: BASELINE config 2 (single 640x480 pair, seed 1000, 20
iterations, point-to-plane); `--pairs P` batches P pairs per GPU (config 3 = 64).  For N>1 every
rank processes its own pairs (seed 1000 + rank*P + i) -- the path has no data-path collective --
and the SE(3) pose records are all-gathered over RCCL once per step (weak scaling).
`--mode dense` is BASELINE config 5: ONE pair whose source rows are sharded over the ranks, with a
29-double all-reduce per iteration (strong scaling).

Prints ONE JSON line on rank 0.
  roofline            the dominant kernel of the measured (default, tile-pruned) path: k_nn_tiles_acc.
                      It streams each array once per iteration, so it is accounted against HBM:
                      achieved = algorithmic bytes per launch (12 B src xyz + 4 B idx per
                      valid source point, 12 B xyz + 12 B normal per valid target point) / mean launch
                      duration measured with HIP events on the launch stream.
  roofline_bruteforce the north-star algorithm (full brute-force scan, SLAM3D_NN_BRUTE_VALU) measured in
                      the same process on the same pair: 8*n_src*n_tgt flop per launch vs the fp32 peak.
  cpu_baseline        the CPU oracle (exact kd-tree NN, OpenMP) timed on the host on the same pair.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md:40-41 (vector == f32-MFMA dense peak)
HBM_PEAK_GBPS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md:35 (spec; 6.29 TB/s measured copy)


_FINAL = []      # rank 0's result line, printed after all teardown so that it is the last line on stdout


def _emit_final():
    """RCCL prints a version banner through C stdio; flush that first, then print the one JSON line and flush."""
    if not _FINAL:
        return
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(_FINAL[-1], flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=1, help="frame pairs per GPU per step (config 3: 64)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--estimator", choices=["point2plane", "svd"], default="point2plane")
    ap.add_argument("--nn-mode", type=int, default=0, help="0 auto(tiles) 1 brute-force VALU 2 brute-force MFMA 3 tiles")
    ap.add_argument("--mode", choices=["batch", "dense", "seg", "voxel"], default="batch",
                    help="seg: row f-2, batched RANSAC plane segmentation of --pairs frames per GPU; "
                         "voxel: row f-1, PassThrough + VoxelGrid(0.03) of one resident cloud per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bruteforce", action="store_true")
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + --one-device: exercise the N>1 code path with several ranks on ONE GPU (tests only)")
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--force-collective", action="store_true", help="developer knob: initialise RCCL and run the per-step pose all-gather even with one rank")
    ap.add_argument("--no-pipeline", action="store_true", help="one handle, every step fetched before the next is queued")
    ap.add_argument("--in-flight", type=int, default=3, help="steps in flight (handles taking turns, one HIP stream each); 1 = one step at a time")
    return ap.parse_args()


def cpu_baseline_leg(pair, s4, t4, args, gpu_result, gpu_idx):
    """Times the CPU oracle on the same pair and checks the GPU result against it.
    (oracle use is confined to this leg: checker + CPU baseline, never the measured path)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    est = 0 if args.estimator == "point2plane" else 1
    cores = os.cpu_count() or 1
    p_all = O.params(pair.intr, estimator=est, iterations=args.iterations, nn_method=1, threads=0)
    times = []
    ro = None
    for _ in range(9):
        t0 = time.perf_counter()
        ro = O.icp(s4, t4, p_all, trace=True)
        times.append(time.perf_counter() - t0)
    t_all = statistics.median(times)
    # PCL's ICP is single-threaded: time one thread on a bounded sample (10 iterations, ~1.5 s)
    it1 = min(10, args.iterations)
    p_one = O.params(pair.intr, estimator=est, iterations=it1, nn_method=1, threads=1)
    t0 = time.perf_counter()
    O.icp(s4, t4, p_one, trace=False)
    t_one = time.perf_counter() - t0
    rot, tr = O.pose_error(ro["T_trace"][-1], gpu_result["T_raw"])
    out = {
        "value": args.iterations / t_all, "unit": "ICP iterations/s", "cores": cores, "kind": "port",
        "sample": f"oracle/ (exact kd-tree NN + same estimator, OpenMP on all {cores} hardware threads), 1 pair seed "
                  f"{pair.seed} x {args.iterations} iterations incl. normals + kd-tree build, median of 9 (~4 s on all cores)",
        "single_thread_value": it1 / t_one,
        "single_thread_sample": f"same, 1 thread, {it1} iterations (PCL's own ICP is single-threaded)",
    }
    parity = {
        "rot_err_rad": rot, "trans_err_m": tr,
        "idx_mismatches": int((gpu_idx != ro["idx"]).sum()),
        "T_bit_identical": bool(np.array_equal(ro["T_trace"][-1], gpu_result["T_raw"])),
    }
    return out, parity


def bruteforce_leg(capi, intr, est, d_src_ptr, d_tgt_ptr, local_rank, iterations=4):
    """The north-star algorithm on the same resident pair: every source x every target, distance step as a
    dense contraction on the f32 MFMA pipe (k_nn_mfma); the fp32-VALU scan (k_nn_valu) is timed beside it."""
    out = {}
    for mode, name in ((capi.NN_BRUTE_MFMA, "mfma"), (capi.NN_BRUTE_VALU, "valu")):
        params = capi.default_params(intr, estimator=est, iterations=iterations, max_batch=1, device=local_rank, nn_mode=mode)
        with capi.IcpHandle(params) as h:
            h.set_clouds_device(0, d_src_ptr, d_tgt_ptr)
            h.set_profiling(True)
            h.run(1)
            h.fetch_results(1)                       # warm-up
            h.run(1)
            r = h.fetch_results(1)[0]
            ms = float(np.mean(h.get_iteration_timings()[1:]))    # iteration 0 has no previous match to bound the filter
        out[name] = (ms, 8.0 * r["n_src"] * r["n_tgt"])
    ms, flops = out["mfma"]
    ach = flops / (ms * 1e-3) / 1e12
    vms, _ = out["valu"]
    return {"kernel": "k_nn_mfma (full brute-force scan: v_mfma_f32_16x16x4_f32 distance contraction as a conservative "
                      "filter + exact fp32 re-evaluation of flagged pairs; bit-identical results)",
            "bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS,
            "traffic": None, "launch_ms": ms, "flops_per_launch": flops, "iterations_per_s_if_used": 1e3 / ms,
            "valu_kernel": {"kernel": "k_nn_valu (same scan on the fp32 VALU)", "launch_ms": vms,
                            "achieved": flops / (vms * 1e-3) / 1e12, "frac": flops / (vms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}}


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the driver compares it)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "ICP iterations/sec on 640\u00d7480 clouds; SE(3) pose error vs PCL ref"


def committed_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (not a live measurement)."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return d["k_nn_tiles_acc"]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), d["k_nn_tiles_acc"]
    except Exception:
        return None, None, {}


def seg_mode(args, torch, dist, capi, synth, world, rank, local_rank, dev):
    """Row f-2: plane segmentation of F = --pairs resident frames per GPU and step."""
    F = args.pairs
    seeds = [args.seed0 + rank * F + i for i in range(F)]
    frames = [synth.make_pair(s, args.width, args.height) for s in seeds]
    intr = frames[0].intr
    host = [synth.backproject_numpy(p.depth_src, intr).reshape(-1, 4) for p in frames]
    d = torch.from_numpy(np.stack(host)).to(dev)
    N = args.width * args.height
    d_lab = torch.zeros((F, N), dtype=torch.int32, device=dev)
    h = capi.IcpHandle(capi.default_params(intr, max_batch=F, device=local_rank))
    sp = h.seg_params(seed=args.seed0)
    ptrs = [d.data_ptr() + i * N * 16 for i in range(F)]
    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out_planes = None
    for _ in range(args.warmup):
        h.segment_planes_device(ptrs, sp, d_lab.data_ptr(), stream)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_planes = h.segment_planes_device(ptrs, sp, d_lab.data_ptr(), stream)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    rounds = [len(p) for p in out_planes]
    # point passes: init + per round (count, moments, label); each reads 16 B cloud + 4 B label per pixel
    passes = sum(1 + 3 * min(sp.max_planes, r + 1) for r in rounds)
    alg_bytes = passes * 20 * N
    value = world * F * args.steps / elapsed
    out = {
        "metric": f"plane segmentations/sec on {args.width}x{args.height} organized clouds", "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+int64", "data": "synthetic",
        "config": {"workload": f"row f-2: {F} frame(s) per GPU, <= {sp.max_planes} planes, {sp.hypotheses} hypotheses per round, "
                               f"threshold {sp.distance_threshold:.2f} m, plane_percent {sp.plane_percent:.1f}",
                   "frames_per_gpu": F, "planes_found": rounds[:8]},
        "roofline": {"kernel": "whole launch sequence (k_seg_init + rounds x {hyp, count, moments, refine, label})",
                     "bound": "hbm", "achieved": alg_bytes / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_step": alg_bytes,
                     "note": "17 dependent small launches per call; launch latency, not bandwidth, bounds one frame"},
    }
    if rank == 0:
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            t1 = time.perf_counter()
            po, lo = O.segment_planes(host[0], seed=args.seed0)
            dt = time.perf_counter() - t1
            lab0 = d_lab[0].cpu().numpy()
            out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "oracle/seg_oracle.c, frame 0, single thread"}
            out["parity_vs_oracle"] = {"labels_equal": bool(np.array_equal(lab0, lo)),
                                       "coeff_equal": bool(all(np.array_equal(a["coeff"], b["coeff"]) for a, b in zip(out_planes[0], po)))}
        _FINAL.append(json.dumps(out))
    h.close()


def voxel_mode(args, torch, dist, capi, synth, world, rank, local_rank, dev):
    """Row f-1: PassThrough + VoxelGrid(grid_leaf 0.03) of one resident 16-byte-record cloud per step."""
    pr = synth.make_pair(args.seed0 + rank, args.width, args.height)
    c = synth.backproject_numpy(pr.depth_src, pr.intr).reshape(-1, 4).copy()
    c[:, 3] = np.random.default_rng(args.seed0 + rank).integers(0, 2 ** 32, c.shape[0], dtype=np.uint64).astype(np.uint32).view(np.float32)
    n = c.shape[0]
    d = torch.from_numpy(c).to(dev)
    out_d = torch.zeros_like(d)
    h = capi.IcpHandle(capi.default_params(pr.intr, max_batch=1, device=local_rank))
    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    m = 0
    for _ in range(args.warmup):
        m = h.voxel_grid_device(d.data_ptr(), n, out_d.data_ptr(), 0.03, stream)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m = h.voxel_grid_device(d.data_ptr(), n, out_d.data_ptr(), 0.03, stream)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    alg_bytes = 16 * n + 16 * m
    per = elapsed / args.steps
    out = {
        "metric": f"voxel-grid down-samplings/sec of {args.width}x{args.height} clouds", "value": world * args.steps / elapsed,
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * per,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64 fixed point", "data": "synthetic",
        "config": {"workload": f"row f-1: PassThrough z<=7 + VoxelGrid leaf 0.03 on {n} records of 16 B -> {m} voxels", "points": n, "voxels": m},
        "roofline": {"kernel": "whole launch sequence (table clear, k_voxel_insert, k_voxel_compact, k_voxel_rank)", "bound": "hbm",
                     "achieved": alg_bytes / per / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_bytes / per / 1e9 / HBM_PEAK_GBPS,
                     "traffic": None, "algorithmic_bytes_per_step": alg_bytes,
                     "note": "algorithmic bytes = records in + records out; the hash table (52 B x 2^20 slots) is cleared and scanned "
                             "every call, which is the real traffic"},
    }
    if rank == 0:
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            t1 = time.perf_counter()
            want = O.voxel_grid(c, 0.03, 7.0)
            dt = time.perf_counter() - t1
            got = out_d[:m].cpu().numpy()
            out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "oracle/voxel_oracle.c (qsort by voxel key), same cloud, single thread"}
            out["parity_vs_oracle"] = {"bit_identical": bool(got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)))}
        _FINAL.append(json.dumps(out))
    h.close()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from slam3d_gx_amd import capi, dense, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the ICP path)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    host_comm = args.dist_backend == "gloo"       # exchange buffers on the host (RCCL needs one GPU per rank)
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if host_comm:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.mode in ("seg", "voxel"):
        (seg_mode if args.mode == "seg" else voxel_mode)(args, torch, dist, capi, synth, world, rank, local_rank, dev)
        if dist.is_initialized():
            dist.destroy_process_group()
        _emit_final()
        return
    is_dense = args.mode == "dense"
    P = 1 if is_dense else args.pairs
    est = capi.EST_POINT2PLANE if args.estimator == "point2plane" else capi.EST_SVD
    # ---- synthetic inputs, uploaded once: the timed region starts with clouds resident in HBM
    seeds = [args.seed0] if is_dense else [args.seed0 + rank * P + i for i in range(P)]
    pairs = [synth.make_pair(s, args.width, args.height) for s in seeds]
    intr = pairs[0].intr
    src_host = [synth.backproject_numpy(p.depth_src, intr) for p in pairs]
    tgt_host = [synth.backproject_numpy(p.depth_tgt, intr) for p in pairs]
    d_src = torch.from_numpy(np.stack(src_host)).to(dev)
    d_tgt = torch.from_numpy(np.stack(tgt_host)).to(dev)
    params = capi.default_params(intr, estimator=est, iterations=args.iterations, max_batch=P,
                                 device=local_rank, nn_mode=args.nn_mode)
    # --in-flight handles (each with its own HIP stream) on the same resident inputs take turns (batch mode): the next
    # steps are queued before step k's poses are fetched (fetch waits for its own run's end event only), so
    # consecutive steps overlap on the GPU and the host round trip is hidden.  `single_step_latency_ms` reports the
    # un-overlapped time of one step next to it.
    handles = [capi.IcpHandle(params) for _ in range(1 if (is_dense or args.no_pipeline) else max(1, min(8, args.in_flight)))]
    h = handles[0]
    rec_bytes = 4 * args.width * args.height * 4
    for hh in handles:
        for i in range(P):
            hh.set_clouds_device(i, d_src.data_ptr() + i * rec_bytes, d_tgt.data_ptr() + i * rec_bytes)
    stream = torch.cuda.current_stream().cuda_stream
    table = {}
    d_sums = torch.zeros(29, dtype=torch.int64, device=dev)       # dense mode: the per-iteration exchange buffer
    gatherer = (shard.PoseGatherer(world * P, device=None if host_comm else dev, force=args.force_collective)
                if ((world > 1 or args.force_collective) and not is_dense) else None)
    host_allreduce = dense.allreduce_sum_torch(None) if (host_comm and world > 1) else None

    def step():
        if is_dense:      # one exchange per iteration: 29-double all-reduce (RCCL), device resident
            if host_allreduce:
                return [dense.dense_align(h, world, rank, None, host_allreduce, stream)]
            return [dense.dense_align_device(h, world, rank, d_sums, None, stream)]
        h.run(P, None, stream)
        return finish(h)

    def finish(hh):
        res = hh.fetch_results(P)
        if gatherer:      # RCCL all-gather of the 160-byte pose records (T, norm, inliers, status, rmse), one per
            gatherer.submit(shard.pack_records(res))     # step, overlapping the next step's kernels
            if len(gatherer.pending) > 1:
                table["poses"] = gatherer.collect()
        return res

    def run_steps(n):
        """n steps, software-pipelined over the handles; every step's poses are fetched (and gathered) inside"""
        if len(handles) == 1:
            out = None
            for _ in range(n):
                out = step()
            return out
        out, nh = None, len(handles)
        for k in range(n):
            handles[k % nh].run(P, None, stream)
            if k >= nh - 1:
                out = finish(handles[(k - (nh - 1)) % nh])
        for k in range(max(0, n - (nh - 1)), n):
            out = finish(handles[k % nh])
        return out

    def drain():
        while gatherer and gatherer.pending:
            table["poses"] = gatherer.collect()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    drain()
    nn_ms, tot_ms, pre_ms = [], [], []
    fence()
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    h_last = handles[(args.steps - 1) % len(handles)]
    drain()                 # the last step's pose table is on every rank before the clock stops
    fence()
    elapsed = time.perf_counter() - t0
    latency_ms = None
    if not is_dense:
        nl = min(args.steps, 10)
        fence()
        tl = time.perf_counter()
        for _ in range(nl):
            step()
        drain()
        fence()
        latency_ms = 1e3 * (time.perf_counter() - tl) / nl
        # kernel durations for the roofline: the same K steps again with per-launch HIP events on the launch
        # stream (kept out of the timed region because every event record serialises the stream for ~6 us)
        h.set_profiling(True)
        for _ in range(args.steps):
            step()
            tm = h.get_timings()
            nn_ms.append(tm["nn_ms"]); tot_ms.append(tm["total_ms"]); pre_ms.append(tm["preprocess_ms"])
        drain()
        fence()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_comm else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_iters = (1 if is_dense else world * P) * args.iterations * args.steps
    value = total_iters / elapsed
    size_tag = f"{args.width}x{args.height}"
    out = {
        "metric": baseline_metric() if size_tag == "640x480" else f"ICP iterations/sec on {size_tag} clouds; SE(3) pose error vs PCL ref",
        "value": value, "unit": "ICP iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong" if is_dense else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (f"BASELINE config 5: one {size_tag} pair, source rows sharded over {world} GPU(s), "
                         f"{args.iterations} iterations, {args.estimator}, 232-byte all-reduce per iteration" if is_dense else
                         f"BASELINE config {'2' if P == 1 else '3'}: {P} frame pair(s) per GPU of {size_tag}, "
                         f"{args.iterations} ICP iterations, {args.estimator}, exact NN (tile-pruned brute force), "
                         f"seeds {seeds[0]}..{seeds[-1]}"),
            "pairs_per_gpu": P, "iterations": args.iterations, "estimator": args.estimator,
            "step_pipelining": (f"{len(handles)} handles, each on its own HIP stream, take turns: the following steps are queued before "
                                f"step k's poses are fetched, so {len(handles)} consecutive steps overlap on the GPU; every step's poses "
                                "reach the host inside the timed region") if len(handles) > 1 else "none",
            "nn_mode": {0: "auto(tiles)", 1: "brute_valu", 2: "brute_mfma", 3: "tiles"}.get(args.nn_mode, str(args.nn_mode)),
            "n_src": [r["n_src"] for r in res][:4], "n_tgt": [r["n_tgt"] for r in res][:4],
            "parallelism": (f"source rows over {world} rank(s), RCCL all-reduce of 29 doubles per iteration" if is_dense else
                            f"pairs sharded one process per GPU x{world}, one RCCL all-gather of pose records per step"),
        },
        "status": [r["status"] for r in res][:8],
    }
    if latency_ms is not None:
        out["single_step_latency_ms"] = latency_ms      # one step at a time (no overlap between steps), same inputs
    if not is_dense:
        # ---- roofline of the dominant kernel: one launch = one ICP iteration over the P resident pairs
        launch_ms = statistics.mean(nn_ms) / max(args.iterations, 1)
        alg_bytes = sum((12 + 4) * r["n_src"] + (12 + (12 if est == 0 else 0)) * r["n_tgt"] for r in res)
        flops = sum(8.0 * r["n_src"] * r["n_tgt"] for r in res)
        if args.nn_mode in (capi.NN_BRUTE_VALU, capi.NN_BRUTE_MFMA):
            ach = flops / (launch_ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": "k_nn_mfma" if args.nn_mode == capi.NN_BRUTE_MFMA else "k_nn_valu (full brute-force scan)",
                               "bound": "mfma", "achieved": ach,
                               "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS, "traffic": None,
                               "launch_ms": launch_ms, "flops_per_launch": flops}
        else:
            ach = alg_bytes / (launch_ms * 1e-3) / 1e9
            traffic, src, prof = committed_traffic() if (P == 1 and size_tag == "640x480" and est == 0) else (None, None, {})
            out["roofline"] = {
                "kernel": "k_nn_tiles_acc (exact tile-pruned NN + fused normal-equation accumulation)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": src, "launch_ms": launch_ms,
                "launch_ms_source": "HIP events around every k_nn_tiles_acc launch, second pass over the same K steps",
                "algorithmic_bytes_per_launch": alg_bytes,
                "valu_issue_floor_us": prof.get("valu_issue_floor_us"),      # from the committed PMC profile: the bound that applies
                "valu_instructions_per_wave": prof.get("valu_instructions_per_wave"),
                "valu_issue_frac": (prof.get("valu_issue_floor_us") / (launch_ms * 1e3)) if prof.get("valu_issue_floor_us") else None,
                "equivalent_bruteforce_tflops": flops / (launch_ms * 1e-3) / 1e12,
                "note": ("streaming accounting (each array once per iteration); the kernel is VALU-issue/latency bound, "
                         "not HBM bound -- see DESIGN.md section 6; equivalent_bruteforce_tflops = flops a full scan "
                         "would need / this launch time (exceeds the 157.3 TF peak because >99 % of the pairs are proven "
                         "irrelevant by bounding boxes, not evaluated)"),
            }
        out["kernel_ms_per_step"] = {"preprocess": statistics.mean(pre_ms), "nn": statistics.mean(nn_ms),
                                     "total": statistics.mean(tot_ms)}
        out["nn_ms_per_iteration"] = [round(float(x), 4) for x in h.get_iteration_timings()]
    if rank == 0:
        if world == 1 and not is_dense and not args.no_bruteforce and args.nn_mode in (capi.NN_AUTO, capi.NN_TILES):
            out["roofline_bruteforce"] = bruteforce_leg(capi, intr, est, d_src.data_ptr(), d_tgt.data_ptr(), local_rank)
        if not args.no_cpu_baseline and world == 1 and not is_dense:
            idx, _ = h.get_correspondences(0)
            cb, parity = cpu_baseline_leg(pairs[0], src_host[0], tgt_host[0], args, res[0], idx)
            out["cpu_baseline"] = cb
            out["parity_vs_oracle"] = parity
        # the flat report SURVEY.md 8(d) lists, assembled from the objects above
        rb, rf, cb_, pv = out.get("roofline_bruteforce", {}), out.get("roofline", {}), out.get("cpu_baseline", {}), out.get("parity_vs_oracle", {})
        out["survey_8d"] = {
            "gpus": world, "pairs": (1 if is_dense else world * P), "iters": args.iterations, "wall_s": elapsed,
            "icp_iters_per_s": value, "nn_tflops": rb.get("achieved"), "nn_frac_fp32_peak": rb.get("frac"),
            "hbm_GBps": rf.get("achieved") if rf.get("unit") == "GB/s" else None,
            "hbm_frac_peak": rf.get("frac") if rf.get("unit") == "GB/s" else None,
            "cpu_B_iters_per_s": cb_.get("value"), "cpu_B_1thread_iters_per_s": cb_.get("single_thread_value"), "cores": cb_.get("cores"),
            "max_rot_err": pv.get("rot_err_rad"), "max_trans_err": pv.get("trans_err_m"), "idx_mismatches": pv.get("idx_mismatches"),
        }
        _FINAL.append(json.dumps(out))
    for hh in handles:
        hh.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    _emit_final()


if __name__ == "__main__":
    main()

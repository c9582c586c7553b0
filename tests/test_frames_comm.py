"""GPU tests of the resident-frame store, per-iteration index parity, the RCCL-in-C paths and the full-size configs.

  * frames: a frame's normals / tiles are built once and shared by every pair that references it -- the results must
    be bit-identical to the slot-wise calls (loop-closure candidates share one target frame,
    src/GraphicEnd.cpp:685-762; the keyframe of run() stays the source, :168)
  * "index parity = memcmp of int32 idx[N] per iteration"
  * config 5 (dense) through slam3d_icp_dense_run with a real RCCL communicator (one rank on this
    box; the collective is forced so that ncclAllReduce really runs on the handle's stream), pose gather over RCCL
  * BASELINE configs 3 (full B=64) and 5 (1280x960, all 20 iterations)
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from slam3d_gx_amd import capi, synth

pytestmark = pytest.mark.gpu


def _pair(seed, w=640, h=480, **kw):
    pr = synth.make_pair(seed, w, h, **kw)
    return pr, synth.backproject_numpy(pr.depth_src, pr.intr), synth.backproject_numpy(pr.depth_tgt, pr.intr)


def test_frames_shared_by_many_pairs_equal_slotwise_runs(gpu_lib):
    """8 'loop-closure candidates' (distinct sources) against ONE target frame, through the frame API with depth
    uploads; then the same pairs slot by slot.  Every iterate and the last correspondences must agree bit for bit."""
    w, h, B = 320, 240, 8
    prs = [synth.make_pair(100 + i, w, h) for i in range(B)]
    intr = prs[0].intr
    tgt_depth = prs[0].depth_tgt
    params = capi.default_params(intr, iterations=6, max_batch=B, extra_frames=B + 1)
    with capi.IcpHandle(params) as hd:
        f0 = hd.first_free_frame()
        assert hd.frame_count() == 2 * B + B + 1
        hd.frame_set_depth_host(f0, tgt_depth)                       # the shared target (the new keyframe)
        for i, pr in enumerate(prs):
            hd.frame_set_depth_host(f0 + 1 + i, pr.depth_src)
            hd.set_pair(i, f0 + 1 + i, f0)
        hd.run(B)
        res = hd.fetch_results(B)
        tr = [hd.get_trace(i)[0] for i in range(B)]
        idx = [hd.get_correspondences(i)[0] for i in range(B)]
        # second run on the same frames: everything cached, same bits
        hd.run(B)
        res2 = hd.fetch_results(B)
        # replace ONE source frame: only that pair changes
        hd.frame_set_depth_host(f0 + 1, prs[1].depth_src)
        hd.run(B)
        res3 = hd.fetch_results(B)
    for a, b in zip(res, res2):
        assert np.array_equal(a["T_raw"], b["T_raw"]) and a["inliers"] == b["inliers"]
    assert np.array_equal(res3[0]["T_raw"], res[1]["T_raw"])        # pair 0 now has pair 1's source
    assert np.array_equal(res3[2]["T_raw"], res[2]["T_raw"])
    with capi.IcpHandle(capi.default_params(intr, iterations=6, max_batch=1)) as hs:
        for i, pr in enumerate(prs):
            r = hs.align_depth_batch([pr.depth_src], [tgt_depth])[0]
            assert np.array_equal(r["T_raw"], res[i]["T_raw"]), i
            assert r["inliers"] == res[i]["inliers"] and r["n_src"] == res[i]["n_src"] and r["n_tgt"] == res[i]["n_tgt"]
            assert np.array_equal(hs.get_trace(0)[0], tr[i])
            assert np.array_equal(hs.get_correspondences(0)[0], idx[i])
    # and against the oracle for two of them
    p = O.params(intr, iterations=6, nn_method=1)
    t4 = O.backproject(tgt_depth, p)
    for i in (0, 5):
        ro = O.icp(O.backproject(prs[i].depth_src, p), t4, p)
        assert np.array_equal(ro["T_trace"], tr[i]) and np.array_equal(ro["idx"], idx[i])


def test_borrowed_device_frame_must_be_invalidated_after_an_in_place_update(gpu_lib):
    """frames are preprocessed once per SET.  A borrowed device buffer rewritten in place keeps its old normals and
    tiles until slam3d_icp_frame_invalidate (or a new set) says otherwise -- documented in the header, checked here."""
    import torch
    pa, sa4, ta4 = _pair(3101, 320, 240)
    pb, sb4, tb4 = _pair(3102, 320, 240)
    params = capi.default_params(pa.intr, iterations=6)
    with capi.IcpHandle(params) as h:
        want_a = h.align(sa4, ta4)["T_raw"].copy()
        want_b = h.align(sb4, tb4)["T_raw"].copy()
    d_s, d_t = torch.from_numpy(sa4).to("cuda:0"), torch.from_numpy(ta4).to("cuda:0")
    with capi.IcpHandle(params) as h:
        h.frame_set_cloud_device(0, d_s.data_ptr()); h.frame_set_cloud_device(1, d_t.data_ptr()); h.set_pair(0, 0, 1)
        h.run(1)
        assert np.array_equal(h.fetch_results(1)[0]["T_raw"], want_a)
        d_s.copy_(torch.from_numpy(sb4)); d_t.copy_(torch.from_numpy(tb4))          # the caller rewrites both buffers in place
        torch.cuda.synchronize()
        h.frame_invalidate(0); h.frame_invalidate(1)
        h.run(1)
        assert np.array_equal(h.fetch_results(1)[0]["T_raw"], want_b)
        with pytest.raises(capi.Slam3dError):
            h.frame_invalidate(5)                                                    # never set


def test_frame_api_rejects_bad_use(gpu_lib):
    pr = synth.make_pair(3, 160, 120)
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=2, max_batch=2, extra_frames=1)) as hd:
        with pytest.raises(capi.Slam3dError) as e:
            hd.frame_set_depth_host(5, pr.depth_src)                 # ids 0..4 exist
        assert e.value.code == -1
        with pytest.raises(capi.Slam3dError) as e:
            hd.set_pair(2, 0, 1)
        assert e.value.code == -1
        hd.set_pair(0, 4, 1)
        with pytest.raises(capi.Slam3dError) as e:
            hd.run(1)                                                # frames never set
        assert e.value.code == -5
        hd.frame_set_depth_host(4, pr.depth_src)
        hd.frame_set_depth_host(1, pr.depth_tgt)
        hd.run(1)
        r = hd.fetch_results(1)[0]
        assert r["status"] == 0 and r["inliers"] > 1000


@pytest.mark.parametrize("estimator", [0, 1])
def test_per_iteration_index_parity_config2(gpu_lib, estimator):
    """memcmp of idx[N] at EVERY iteration against the oracle (one exact NN pass per oracle iterate)."""
    pr, s4, t4 = _pair(1000)
    iters = 20
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=iters, estimator=estimator)) as hd:
        hd.set_corr_trace(True)
        r = hd.align(s4, t4)
        Tt, _ = hd.get_trace(0)
        per_it = [hd.get_correspondences_at(k) for k in range(iters)]
        last, _ = hd.get_correspondences(0)
        # the production configuration (graph replay, no trace) gives the same last iteration
        hd.set_corr_trace(False)
        r2 = hd.align(s4, t4)
        assert np.array_equal(r2["T_raw"], r["T_raw"]) and np.array_equal(hd.get_correspondences(0)[0], last)
        with pytest.raises(capi.Slam3dError):
            hd.get_correspondences_at(0)                             # that run was not traced
    assert np.array_equal(per_it[-1], last)
    p = O.params(pr.intr, iterations=iters, estimator=estimator, nn_method=1)
    ro = O.icp(s4, t4, p)
    assert np.array_equal(ro["T_trace"], Tt)
    for k in range(iters):
        want, _, _ = O.nn_once(s4, t4, p, T=ro["T_trace"][k], use_normals=(estimator == 0),
                               coarse=(k < p.coarse_iterations and k < iters - 1))           # spec S4c
        assert np.array_equal(want, per_it[k]), f"iteration {k}: {(want != per_it[k]).sum()} indices differ"


def _comm_one_rank(device=0):
    uid = capi.comm_unique_id()
    assert len(uid) == 128
    return capi.Comm(uid, 0, 1, device)


@pytest.mark.parametrize("head", ["1", "0"])
def test_dense_run_through_rccl_allreduce_is_bit_identical(gpu_lib, monkeypatch, head):
    """Config 5's loop inside the library.  head = 1 (default): launch k accumulates into set k, ONE ncclAllReduce of that set
    in place on the handle's stream, the head of launch k+1 solves; head = 0: partial -> ncclAllReduce of 29 words -> update.
    One rank on this box, collective forced, so RCCL really runs between the kernels; bit-identical to the unsharded batch
    run and to the host-synchronous dense loop."""
    monkeypatch.setenv("SLAM3D_HEAD_SOLVE", head)
    pr, s4, t4 = _pair(2001, 320, 240)
    params = capi.default_params(pr.intr, iterations=10)
    with capi.IcpHandle(params) as hd:
        want = hd.align(s4, t4)
        want_T = hd.get_trace(0)[0]
    comm = _comm_one_rank()
    try:
        monkeypatch.setenv("SLAM3D_DENSE_FORCE_COLLECTIVE", "1")
        with capi.IcpHandle(params) as hd:
            hd.set_clouds_host(0, s4, t4)
            got = hd.dense_run(comm)
            assert np.array_equal(hd.get_trace(0)[0], want_T)
            got2 = hd.dense_run(comm)                                # again on cached frames
        monkeypatch.delenv("SLAM3D_DENSE_FORCE_COLLECTIVE")
        with capi.IcpHandle(params) as hd:
            hd.set_clouds_host(0, s4, t4)
            got3 = hd.dense_run(None)                                # no communicator: one rank, no collective
    finally:
        comm.close()
    for g in (got, got2, got3):
        assert np.array_equal(g["T_raw"], want["T_raw"]) and g["inliers"] == want["inliers"] and g["status"] == want["status"]
        assert g["n_src"] == want["n_src"] and g["n_tgt"] == want["n_tgt"] and g["rmse"] == want["rmse"]


def test_pose_gather_over_rccl_one_rank(gpu_lib):
    comm = _comm_one_rank()
    try:
        rng = np.random.default_rng(1)
        mk = lambda k: [dict(T=rng.normal(size=(4, 4)), norm=float(k), inliers=100 + k, status=k % 3, rmse=0.5 * k) for _ in range(5)]
        a, b = mk(1), mk(2)
        comm.gather_submit(a)
        comm.gather_submit(b)                                        # two in flight
        with pytest.raises(capi.Slam3dError):
            comm.gather_submit(a)                                    # a third is refused
        ga, gb = comm.gather_collect(5), comm.gather_collect(5)
        for src, got in ((a, ga), (b, gb)):
            assert len(got) == 5
            for x, y in zip(src, got):
                assert np.array_equal(x["T"], y["T"]) and x["norm"] == y["norm"] and x["inliers"] == y["inliers"]
                assert x["status"] == y["status"] and x["rmse"] == y["rmse"]
        big = mk(3) * 13                                             # staging grows
        assert len(comm.gather(big)) == 65
    finally:
        comm.close()


def test_torch_dense_path_on_explicit_side_stream(gpu_lib):
    """The Python form of the dense loop (dense.dense_align_device): it must run on an explicit non-default torch
    stream that is both current (what ProcessGroupNCCL orders against) and the launch stream."""
    import torch
    import torch.distributed as dist
    from slam3d_gx_amd import dense
    pr, s4, t4 = _pair(2002, 320, 240)
    params = capi.default_params(pr.intr, iterations=8)
    with capi.IcpHandle(params) as hd:
        want = hd.align(s4, t4)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        d_sums = torch.zeros(36, dtype=torch.int64, device=dev)
        with capi.IcpHandle(params) as hd:
            hd.set_clouds_host(0, s4, t4)
            got = dense.dense_align_device(hd, 1, 0, d_sums, force_collective=True)
            with pytest.raises(ValueError):
                dense.dense_align_device(hd, 2, 0, d_sums, stream=0)      # several ranks on the null stream: refused
    finally:
        dist.destroy_process_group()
    assert np.array_equal(got["T_raw"], want["T_raw"]) and got["inliers"] == want["inliers"]


def test_config3_full_batch_of_64(gpu_lib):
    """BASELINE config 3 at its full size: 64 pairs of 640x480 (seeds 1000..1063) in one launch sequence.  Every pose
    within the stated bound of T_gt, four sampled pairs bit-identical to the oracle."""
    B = 64
    prs = [synth.make_pair(1000 + i) for i in range(B)]
    intr = prs[0].intr
    with capi.IcpHandle(capi.default_params(intr, iterations=20, max_batch=B)) as hd:
        res = hd.align_depth_batch([p.depth_src for p in prs], [p.depth_tgt for p in prs])
        traces = {i: hd.get_trace(i)[0] for i in (0, 21, 42, 63)}
        idxs = {i: hd.get_correspondences(i)[0] for i in (0, 21, 42, 63)}
    for i, (r, pr) in enumerate(zip(res, prs)):
        rot, tr = O.pose_error(pr.T_gt, r["T_raw"])
        assert r["status"] == 0 and rot < 5e-3 and tr < 2e-2, (i, rot, tr)      # noisy, quantised depth: centimetre-level vs T_gt
    p = O.params(intr, iterations=20, nn_method=1)
    for i in (0, 21, 42, 63):
        ro = O.icp(O.backproject(prs[i].depth_src, p), O.backproject(prs[i].depth_tgt, p), p)
        assert np.array_equal(ro["T_trace"], traces[i]) and np.array_equal(ro["idx"], idxs[i]), i
        assert ro["inliers"] == res[i]["inliers"]


def test_config5_dense_1280x960_all_20_iterations(gpu_lib):
    """BASELINE config 5 on one GPU through slam3d_icp_dense_run: 1280x960 (seed 2000), 20 iterations; bit-identical
    to the oracle (kd-tree), pose error vs T_gt within the stated bound."""
    pr, s4, t4 = _pair(2000, 1280, 960)
    with capi.IcpHandle(capi.default_params(pr.intr, iterations=20)) as hd:
        hd.set_clouds_host(0, s4, t4)
        got = hd.dense_run(None)
        Tt, St = hd.get_trace(0)
        idx, _ = hd.get_correspondences(0)
    ro = O.icp(s4, t4, O.params(pr.intr, iterations=20, nn_method=1))
    assert np.array_equal(ro["T_trace"], Tt) and np.array_equal(ro["sums_trace"], St)
    assert np.array_equal(ro["idx"], idx)
    rot, tr = O.pose_error(ro["T_trace"][-1], got["T_raw"])
    assert rot <= 1e-4 and tr <= 1e-4
    rot, tr = O.pose_error(pr.T_gt, got["T_raw"])
    assert got["status"] == 0 and rot < 5e-3 and tr < 5e-3


def test_failed_solve_is_never_reported_ok(gpu_lib):
    """a solve that fails (here: every normal is the same -> rank-deficient 6x6, but the points still
    match) must not come back as OK with T = T_init.  Oracle and HIP agree on the status."""
    w, h = 160, 120
    intr = synth.Intrinsics.scaled(w, h)
    depth = np.full((h, w), 2000, dtype=np.uint16)                   # a fronto-parallel wall: one normal only
    s4 = synth.backproject_numpy(depth, intr)
    Ti = np.eye(4); Ti[0, 3] = 0.01
    with capi.IcpHandle(capi.default_params(intr, iterations=3)) as hd:
        r = hd.align(s4, s4, Ti)
    ro = O.icp(s4, s4, O.params(intr, iterations=3, nn_method=0), T_init=Ti)
    assert r["status"] == ro["status"] and r["status"] == 3           # DEGENERATE (damped or unsolved), T = Identity
    assert np.array_equal(r["T"], np.eye(4)) and np.array_equal(r["T_raw"], ro["T_trace"][-1])


@pytest.mark.parametrize("estimator", [0, 1])
def test_dense_failure_injection_two_ranks(gpu_lib, tmp_path, estimator):
    """the dense loop's failure protocol with TWO ranks (two processes on this GPU, exchange over gloo through
    slam3d_icp_dense_run_with -- the same library loop as slam3d_icp_dense_run, RCCL needs a GPU per rank).  Healthy: both ranks end
    with the oracle's pose bits.  Rank 1 fails in iteration 3 (slam3d_icp_set_fault_injection): it returns its own error (SLAM3D_E_HIP) after
    completing the remaining exchanges with the failure word set; rank 0 returns SLAM3D_E_COMM -- within the timeout, not a hang.
    estimator 0: the head-solve flow (the accumulator set is exchanged in place); 1: the three-step flow (36 totals + the word)."""
    import json, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr, s4, t4 = _pair(1000, 320, 240)
    ro = O.icp(s4, t4, O.params(pr.intr, iterations=8, nn_method=1, estimator=estimator))
    # -1 healthy; 3: iteration 3 cannot be enqueued; -2: the rank fails before its first iteration (used to leave the peer in its
    # first all-reduce); 1003: behind iteration 3's exchange (the drain must not repeat that exchange); 1007: behind the LAST exchange --
    # every exchange is complete, the healthy rank's pose is valid and it returns it
    for fail_at in (-1, 3, -2, 1003, 1007):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        outs = [str(tmp_path / f"r{r}_{fail_at}.json") for r in range(2)]
        procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "_dense_fail_worker.py"), str(r), "2", port, str(estimator), str(fail_at), outs[r]],
                                  cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        logs = []
        for p_ in procs:
            try:
                logs.append(p_.communicate(timeout=240)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                pytest.fail(f"a rank hung (fail_at={fail_at})")
        assert all(p_.returncode == 0 for p_ in procs), logs
        res = [json.load(open(o)) for o in outs]
        if fail_at == 1007:
            assert res[1]["code"] == -2 and "injected failure" in res[1]["message"], res[1]
            r = res[0]
            assert r["code"] == 0 and r["status"] == ro["status"] and np.array_equal(np.array(r["T"]).reshape(4, 4), ro["T_trace"][-1]), r
        elif fail_at == -1:
            for r in res:
                assert r["code"] == 0 and r["status"] == ro["status"] and r["inliers"] == ro["inliers"]
                assert np.array_equal(np.array(r["T"]).reshape(4, 4), ro["T_trace"][-1])
        else:
            assert res[1]["code"] == -2 and "injected failure" in res[1]["message"], res[1]
            assert res[0]["code"] == -6 and "peer rank" in res[0]["message"], res[0]

"""Python twin of the front-end control flow (test infrastructure): an independent re-statement, in another language, of
GraphicEnd::run / generateKeyFrame / loopClosure / lostRecovery / check / checknearby / findMoreLoops
(src/GraphicEnd.cpp:150-264, 304-351, 685-762, 764-838, 868-947) as host/GraphicEndICP.cpp implements them.  It drives the
same C-ABI through ctypes (the poses come from the HIP library either way), so that what is compared is the CONTROL FLOW:
which pairs are aligned, which results are accepted, which keyframes / edges / log lines come out.  tests/
test_host_frontend.py runs run_SLAM and this twin on the same sequence and requires identical keyframe.txt, lc.txt, lost.txt,
error_of_transform.log and the same edge list in final.g2o.
"""
from __future__ import annotations

import numpy as np

from slam3d_gx_amd import capi

M64 = (1 << 64) - 1


def splitmix_next(state):
    state = (state + 0x9E3779B97F4A7C15) & M64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    z ^= z >> 31
    return state, z


def inv_rigid(T):
    R = np.eye(4)
    R[:3, :3] = T[:3, :3].T
    R[:3, 3] = -(R[:3, :3] @ T[:3, 3])
    return R


class Twin:
    def __init__(self, intr, depth_of, cfg, cloud_of=None):
        """depth_of(frame_index) -> uint16 image; cfg: dict of the parameters.yaml values run_SLAM was given;
        cloud_of(frame_index) -> [n, 4] float32 voxel cloud (icp_cloud: voxel -- multiPnP aligns readimage's point lists)"""
        self.depth_of = depth_of
        self.cloud_of = cloud_of
        self.hl = None
        self.c = dict(max_pos_change=0.25, error_threshold=1.0, lost_frames=10, loop_closure_error=1.5, loop_closure_inliers=30,
                      loop_closure_detection=False, loopclosure_frames=30, loopclosure_seed=1, icp_iterations=20, icp_min_inliers=12,
                      icp_min_inlier_ratio=0.3, icp_max_rmse=0.05, icp_loop_min_inlier_ratio=0.6, icp_loop_max_rmse=0.02, start_index=1)
        self.c.update(cfg)
        self.h = capi.IcpHandle(capi.default_params(intr, iterations=self.c["icp_iterations"], min_inliers=self.c["icp_min_inliers"],
                                                    error_threshold=self.c["error_threshold"], max_batch=1,
                                                    max_plane_residual2=self.c.get("max_plane_residual2", 0.0),
                                                    min_normal_cos=self.c.get("min_normal_cos", 0.0),
                                                    estimator=self.c.get("estimator", 0), plane_flags=self.c.get("plane_flags", 0)))
        if self.c.get("estimator", 0) == capi.EST_PLANE:      # GraphicEndICP hands the library parameters.yaml's plane keys, seed 1
            self.h.set_seg_params(self.h.seg_params(distance_threshold=self.c.get("icp_seg_distance_threshold", 0.04), plane_percent=self.c.get("plane_percent", 0.2),
                                                    max_planes=self.c.get("max_planes", 3), hypotheses=self.c.get("ransac_hypotheses", 64), seed=1))
        if self.c.get("icp_cloud") == "voxel":                # GraphicEndICP's second handle: lists of icp_cloud_max_points, svd
            cap = int(self.c.get("icp_cloud_max_points", 32768))
            self.cap = cap
            self.hl = capi.IcpHandle(capi.default_params(type(intr)(width=cap, height=1), iterations=self.c["icp_iterations"],
                                                         min_inliers=self.c["icp_min_inliers"], error_threshold=self.c["error_threshold"],
                                                         max_batch=1, estimator=capi.EST_SVD))
        self.index = self.c["start_index"]
        self.lost = 0
        self.lc_state = self.c["loopclosure_seed"]
        self.keyframes = []            # dicts: id, frame_index, connect
        self.edges = []                # (from id, to id)
        self.lc_lines, self.lost_lines, self.err_log = [], [], []
        self.kf_pos = np.eye(4)
        self.robot = np.eye(4)
        self.traj = []
        first = dict(id=0, frame_index=self.index, connect=[])
        self.keyframes.append(first)
        self.cur = first
        self.last = self.index         # frame index of _last
        self.present = self.index
        self.traj.append((self.index, self.robot.copy()))
        self.index += 1

    def close(self):
        self.h.close()
        if self.hl is not None:
            self.hl.close()

    def _list_view(self, f):
        c = np.asarray(self.cloud_of(f), dtype=np.float32)
        out = np.zeros((1, len(c), 4), dtype=np.float32)
        out[0, :, :3] = c[:, :3]
        return out

    # ---- multiPnP with the gates of GraphicEndICP::alignOnDevice
    def multi_pnp(self, f1, f2, loop=False, min_inliers=None, T_init=None):
        min_inliers = self.c["icp_min_inliers"] if min_inliers is None else min_inliers
        if self.hl is not None:
            r = self.hl.align(self._list_view(f1), self._list_view(f2), T_init)
        else:
            r = self.h.align_depth_batch([self.depth_of(f1)], [self.depth_of(f2)], None if T_init is None else [T_init])[0]
        ratio = self.c["icp_loop_min_inlier_ratio"] if loop else self.c["icp_min_inlier_ratio"]
        rmse = self.c["icp_loop_max_rmse"] if loop else self.c["icp_max_rmse"]
        good = r["status"] == 0 and r["inliers"] >= min_inliers
        if good and ratio > 0 and r["n_src"] > 0 and r["inliers"] < ratio * r["n_src"]:
            good = False
        if good and rmse > 0 and r["rmse"] > rmse:
            good = False
        T = r["T"] if good else np.eye(4)
        return dict(T=T, norm=r["norm"], inliers=r["inliers"], identity=bool(np.array_equal(T, np.eye(4))))

    def accept_loop(self, r):
        return (not r["identity"]) and r["norm"] <= self.c["loop_closure_error"] and r["inliers"] >= self.c["loop_closure_inliers"]

    def generate_keyframe(self, T, frame_index):
        kf = dict(id=len(self.keyframes), frame_index=frame_index, connect=[])
        self.kf_pos = self.kf_pos @ T
        self.keyframes.append(kf)
        self.edges.append((kf["id"] - 1, kf["id"]))
        self.cur = kf

    def loop_closure(self):
        n = len(self.keyframes)
        if n <= 3:
            return
        cand = [n + i for i in (-3, -4) if n + i >= 0]
        n_adj = len(cand)
        checked = []
        for _ in range(self.c["loopclosure_frames"]):
            self.lc_state, z = splitmix_next(self.lc_state)
            frame = z % (n - 3)
            if frame in checked:
                continue
            checked.append(frame)
            cand.append(frame)
        for k, ci in enumerate(cand):
            r = self.multi_pnp(self.keyframes[ci]["frame_index"], self.cur["frame_index"], True, self.c["loop_closure_inliers"])
            if not self.accept_loop(r):
                continue
            self.edges.append((self.keyframes[ci]["id"], self.cur["id"]))
            if k >= n_adj:
                self.lc_lines.append((self.keyframes[ci]["frame_index"], self.cur["frame_index"], r["norm"], r["inliers"]))
                self.keyframes[-1]["connect"].append(ci)

    def lost_recovery(self):
        kf = dict(id=len(self.keyframes), frame_index=self.index, connect=[])
        self.kf_pos = self.robot.copy()
        self.lost_lines.append((kf["id"], kf["frame_index"]))
        self.keyframes.append(kf)
        self.cur = kf
        for i in range(len(self.keyframes) - 1):
            r = self.multi_pnp(self.keyframes[i]["frame_index"], kf["frame_index"], True, self.c["loop_closure_inliers"])
            if self.accept_loop(r):
                self.edges.append((self.keyframes[i]["id"], kf["id"]))
                kf["connect"].append(i)
        self.lost = 0

    def run(self):
        self.present = self.index
        guess = getattr(self, "guess", None) if self.c.get("icp_motion_model") else None      # GraphicEndICP::_T_guess
        res = self.multi_pnp(self.cur["frame_index"], self.present, T_init=guess)
        self.guess = None
        T = inv_rigid(res["T"])
        if res["identity"]:
            self.err_log.append("9999")
            r = self.multi_pnp(self.last, self.present)
            if r["identity"] or r["inliers"] < self.c["loop_closure_inliers"] or r["norm"] > self.c["loop_closure_error"]:
                self.lost += 1
            else:
                self.lost = 0
                rr = self.multi_pnp(self.cur["frame_index"], self.last)
                self.generate_keyframe(inv_rigid(rr["T"]), self.last)
                self.generate_keyframe(inv_rigid(r["T"]), self.index)
                self.robot = self.kf_pos.copy()
                self.last = self.present
        elif res["norm"] > self.c["max_pos_change"]:
            self.err_log.append(res["norm"])
            self.robot = self.kf_pos @ T
            self.generate_keyframe(T, self.index)
            if self.c["loop_closure_detection"]:
                self.loop_closure()
            self.lost = 0
            self.last = self.present
        else:
            self.err_log.append(res["norm"])
            self.robot = self.kf_pos @ T
            self.lost = 0
            self.last = self.present
            self.guess = res["T"]          # an ordinary tracked frame leaves its pose as the next frame's starting guess
        if self.lost > self.c["lost_frames"]:
            self.lost_recovery()
            self.last = self.present
        self.traj.append((self.index, self.robot.copy()))
        self.index += 1

    # ---- saveFinalResult: findMoreLoops
    def check(self, f1, f2):
        r = self.multi_pnp(self.keyframes[f1]["frame_index"], self.keyframes[f2]["frame_index"], True, self.c["loop_closure_inliers"])
        if not self.accept_loop(r):
            return False
        self.edges.append((self.keyframes[f1]["id"], self.keyframes[f2]["id"]))
        return True

    def checknearby(self, source, target):
        checked = []
        index = target
        while index > 0:
            index -= 1
            if index == source:
                continue
            if self.check(source, index):
                checked.append(index)
            else:
                break
        index = target
        while index < len(self.keyframes) - 1:
            index += 1
            if index == source:
                continue
            if self.check(source, index):
                checked.append(index)
            else:
                break
        return checked

    def find_more_loops(self):
        for i, kf in enumerate(self.keyframes):
            for j in list(kf["connect"]):
                for k in self.checknearby(i, j):
                    self.checknearby(k, i)

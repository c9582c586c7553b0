"""ctypes binding of the C-ABI in include/slam3d_icp.h (libslam3d_icp.so, HIP/gfx950).

Plumbing only: the product is the shared library.  There is NO CPU fallback -- if the
library is missing or no MI355X is visible, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import build as _build

NSUMS = 29      # the derived sums of the trace
NRAW = 36       # the integer Gram totals the dense mode exchanges (SLAM3D_ICP_NRAW)
EST_POINT2PLANE, EST_SVD, EST_PLANE = 0, 1, 2
PLANE_PAIR_GATE, PLANE_ONLY = 1, 2        # slam3d_icp_params.plane_flags
NN_AUTO, NN_BRUTE_VALU, NN_BRUTE_MFMA, NN_TILES = 0, 1, 2, 3

EXPORTED_SYMBOLS = [
    "slam3d_icp_default_params", "slam3d_icp_create", "slam3d_icp_destroy", "slam3d_strerror",
    "slam3d_last_error", "slam3d_icp_abi_version", "slam3d_icp_align", "slam3d_icp_align_batch",
    "slam3d_icp_align_depth_batch", "slam3d_icp_set_clouds_host", "slam3d_icp_set_depth_host",
    "slam3d_icp_set_clouds_device", "slam3d_icp_set_depth_device", "slam3d_icp_run",
    "slam3d_icp_fetch_results", "slam3d_icp_get_correspondences", "slam3d_icp_get_trace",
    "slam3d_icp_get_clouds", "slam3d_icp_set_profiling", "slam3d_icp_get_timings", "slam3d_icp_get_iteration_timings", "slam3d_icp_set_stamping", "slam3d_icp_get_stamps", "slam3d_icp_get_nn_debug", "slam3d_backproject_u16", "slam3d_fit_planes",
    "slam3d_match_planes", "slam3d_voxel_grid", "slam3d_voxel_grid_device", "slam3d_voxel_grid_only", "slam3d_pass_transform", "slam3d_seg_default_params", "slam3d_segment_planes", "slam3d_segment_planes_device",
    "slam3d_icp_dense_set_rows", "slam3d_icp_dense_begin", "slam3d_icp_dense_partial",
    "slam3d_icp_dense_update", "slam3d_icp_dense_finish",
    "slam3d_icp_dense_partial_device", "slam3d_icp_dense_update_device", "slam3d_icp_dense_finish_device",
    "slam3d_icp_frame_count", "slam3d_icp_frame_set_depth_host", "slam3d_icp_frame_set_depth_device",
    "slam3d_icp_frame_set_cloud_host", "slam3d_icp_frame_set_cloud_device", "slam3d_icp_frame_invalidate", "slam3d_icp_set_pair",
    "slam3d_icp_set_corr_trace", "slam3d_icp_get_correspondences_at",
    "slam3d_comm_get_unique_id", "slam3d_comm_init", "slam3d_comm_destroy", "slam3d_comm_rank", "slam3d_comm_world",
    "slam3d_comm_last_error", "slam3d_shard_range", "slam3d_icp_dense_run", "slam3d_pose_gather_submit",
    "slam3d_pose_gather_collect", "slam3d_pose_gather", "slam3d_pose_record_from_result",
    "slam3d_plane_gate", "slam3d_device_count",
    "slam3d_icp_set_seg_params", "slam3d_icp_get_frame_planes", "slam3d_icp_get_plane_assoc", "slam3d_voxel_grid_batch_device", "slam3d_icp_dense_run_with", "slam3d_icp_set_fault_injection", "slam3d_voxel_grid_path_counts",
]
COMM_ID_BYTES = 128


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("depth_factor", C.c_double), ("z_filter", C.c_double),
        ("iterations", C.c_int32), ("max_corr_dist", C.c_double), ("estimator", C.c_int32),
        ("normal_window", C.c_int32), ("normal_min_inliers", C.c_int32), ("normal_inlier_dist", C.c_double),
        ("min_inliers", C.c_int32), ("error_threshold", C.c_double),
        ("max_batch", C.c_int32), ("device", C.c_int32), ("nn_mode", C.c_int32), ("extra_frames", C.c_int32),
        ("max_plane_residual2", C.c_float), ("min_normal_cos", C.c_float),
        ("coarse_iterations", C.c_int32), ("plane_flags", C.c_int32),
    ]


class CloudView(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride_bytes", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class Result(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16), ("norm", C.c_double), ("inliers", C.c_int32), ("status", C.c_int32),
        ("iterations", C.c_int32), ("n_src", C.c_int32), ("n_tgt", C.c_int32), ("_pad", C.c_int32),
        ("rmse", C.c_double), ("T_raw", C.c_double * 16),
    ]

    def as_dict(self):
        return dict(T=np.array(self.T).reshape(4, 4), T_raw=np.array(self.T_raw).reshape(4, 4), norm=self.norm,
                    inliers=self.inliers, status=self.status, iterations=self.iterations, n_src=self.n_src,
                    n_tgt=self.n_tgt, rmse=self.rmse)


class PoseRecord(C.Structure):      # slam3d_pose_record, 160 bytes
    _fields_ = [("T", C.c_double * 16), ("norm", C.c_double), ("inliers", C.c_int32), ("status", C.c_int32),
                ("rmse", C.c_double), ("_pad", C.c_double)]


class Plane(C.Structure):
    _fields_ = [("coeff", C.c_float * 4), ("count", C.c_int32), ("centroid", C.c_float * 3)]


class SegParams(C.Structure):
    _fields_ = [("distance_threshold", C.c_float), ("plane_percent", C.c_float), ("max_planes", C.c_int32),
                ("hypotheses", C.c_int32), ("seed", C.c_uint64)]


class Slam3dError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"slam3d_icp error {code}: {msg}")
        self.code = code


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the in-tree HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    so = path or os.environ.get("SLAM3D_LIB") or _build.LIB        # SLAM3D_LIB: developer knob (kernel variants side by side)
    if not os.path.exists(so):
        raise FileNotFoundError(
            f"{so} not found: build it with `python -m slam3d_gx_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the ICP path.")
    lib = C.CDLL(so)
    lib.slam3d_strerror.restype = C.c_char_p
    lib.slam3d_last_error.restype = C.c_char_p
    lib.slam3d_last_error.argtypes = [C.c_void_p]
    lib.slam3d_icp_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
    lib.slam3d_icp_destroy.argtypes = [C.c_void_p]
    lib.slam3d_icp_destroy.restype = None
    void_fns = ("slam3d_icp_destroy", "slam3d_icp_default_params", "slam3d_seg_default_params", "slam3d_comm_destroy",
                "slam3d_shard_range", "slam3d_pose_record_from_result")
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name in void_fns:
            fn.restype = None
        elif name not in ("slam3d_strerror", "slam3d_last_error", "slam3d_comm_last_error"):
            fn.restype = C.c_int
    lib.slam3d_comm_last_error.restype = C.c_char_p
    lib.slam3d_comm_last_error.argtypes = [C.c_void_p]
    lib.slam3d_comm_destroy.argtypes = [C.c_void_p]
    lib.slam3d_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.slam3d_comm_rank.argtypes = [C.c_void_p]
    lib.slam3d_comm_world.argtypes = [C.c_void_p]
    lib.slam3d_icp_frame_count.argtypes = [C.c_void_p]
    lib.slam3d_icp_default_params.restype = None
    lib.slam3d_seg_default_params.restype = None
    if path is None:
        _lib = lib
    return lib


def match_planes(coeffs1, coeffs2):
    """(n1,4), (n2,4) plane coefficients -> (train_idx[n1], distance[n1]); host code, no device needed"""
    a = np.ascontiguousarray(coeffs1, dtype=np.float32).reshape(-1, 4)
    b = np.ascontiguousarray(coeffs2, dtype=np.float32).reshape(-1, 4)
    pa, pb = (Plane * max(1, a.shape[0]))(), (Plane * max(1, b.shape[0]))()
    for i in range(a.shape[0]):
        pa[i].coeff[:] = a[i].tolist()
    for j in range(b.shape[0]):
        pb[j].coeff[:] = b[j].tolist()
    idx = np.zeros(max(1, a.shape[0]), dtype=np.int32)
    dist = np.zeros(max(1, a.shape[0]), dtype=np.float32)
    rc = load_library().slam3d_match_planes(pa, C.c_int32(a.shape[0]), pb, C.c_int32(b.shape[0]), _vp(idx), _vp(dist))
    if rc:
        raise Slam3dError(rc, "slam3d_match_planes")
    return idx[: a.shape[0]], dist[: a.shape[0]]


def plane_gate(coeffs1, coeffs2, T, max_dist: float = 0.15) -> int:
    """planes of frame 1 carried into frame 2 by T and matched like GraphicEnd::match: number of pairs within max_dist"""
    a = np.ascontiguousarray(coeffs1, dtype=np.float32).reshape(-1, 4)
    b = np.ascontiguousarray(coeffs2, dtype=np.float32).reshape(-1, 4)
    pa, pb = (Plane * max(1, a.shape[0]))(), (Plane * max(1, b.shape[0]))()
    for i in range(a.shape[0]):
        pa[i].coeff[:] = a[i].tolist()
    for j in range(b.shape[0]):
        pb[j].coeff[:] = b[j].tolist()
    Tm = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    n = C.c_int32(0)
    rc = load_library().slam3d_plane_gate(pa, C.c_int32(a.shape[0]), pb, C.c_int32(b.shape[0]), _vp(Tm), C.c_float(max_dist), C.byref(n))
    if rc:
        raise Slam3dError(rc, "slam3d_plane_gate")
    return n.value


def default_params(intr=None, **kw) -> Params:
    p = Params()
    load_library().slam3d_icp_default_params(C.byref(p))
    if intr is not None:
        p.width, p.height = intr.width, intr.height
        p.fx, p.fy, p.cx, p.cy, p.depth_factor = intr.fx, intr.fy, intr.cx, intr.cy, intr.depth_factor
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _vp(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _cloud_view(a: np.ndarray, w: int, h: int) -> CloudView:
    """a: (H, W, k) float32 array with k*4-byte records (k >= 3), C-contiguous."""
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.shape[0] == h and (a.shape[1] == w or (h == 1 and a.shape[1] <= w))
    return CloudView(a.ctypes.data, a.shape[2] * 4, a.shape[1], h)       # (an unorganized handle, h == 1, takes clouds of any width <= w)


class IcpHandle:
    """RAII wrapper of slam3d_icp_handle."""

    def __init__(self, params: Params):
        self.lib = load_library()
        self.params = params
        self._h = C.c_void_p()
        rc = self.lib.slam3d_icp_create(C.byref(params), C.byref(self._h))
        if rc != 0:
            raise Slam3dError(rc, self.lib.slam3d_strerror(rc).decode())
        self.N = params.width * params.height

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.slam3d_icp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int, allow_algorithmic: bool = True):
        if rc < 0 or (rc > 0 and not allow_algorithmic):
            detail = self.lib.slam3d_last_error(self._h).decode()
            raise Slam3dError(rc, self.lib.slam3d_strerror(rc).decode() + (": " + detail if detail else ""))
        return rc

    # ---- one-call API ------------------------------------------------------------------
    def align(self, src: np.ndarray, tgt: np.ndarray, T_init=None) -> dict:
        return self.align_batch([src], [tgt], None if T_init is None else [T_init])[0]

    def align_batch(self, src: Sequence[np.ndarray], tgt: Sequence[np.ndarray], T_init=None) -> list:
        B = len(src)
        W, H = self.params.width, self.params.height
        sv = (CloudView * B)(*[_cloud_view(np.ascontiguousarray(a, dtype=np.float32), W, H) for a in src])
        tv = (CloudView * B)(*[_cloud_view(np.ascontiguousarray(a, dtype=np.float32), W, H) for a in tgt])
        self._keep = (src, tgt)
        Ti = None if T_init is None else np.ascontiguousarray(np.stack(T_init), dtype=np.float64).reshape(B, 16)
        out = (Result * B)()
        self._check(self.lib.slam3d_icp_align_batch(self._h, C.c_int32(B), sv, tv, _vp(Ti), out))
        return [r.as_dict() for r in out]

    def align_depth_batch(self, src_depth: Sequence[np.ndarray], tgt_depth: Sequence[np.ndarray], T_init=None) -> list:
        B = len(src_depth)
        sd = [np.ascontiguousarray(d, dtype=np.uint16) for d in src_depth]
        td = [np.ascontiguousarray(d, dtype=np.uint16) for d in tgt_depth]
        sp = (C.c_void_p * B)(*[d.ctypes.data for d in sd])
        tp = (C.c_void_p * B)(*[d.ctypes.data for d in td])
        Ti = None if T_init is None else np.ascontiguousarray(np.stack(T_init), dtype=np.float64).reshape(B, 16)
        out = (Result * B)()
        self._check(self.lib.slam3d_icp_align_depth_batch(self._h, C.c_int32(B), sp, tp, _vp(Ti), out))
        return [r.as_dict() for r in out]

    # ---- staged API --------------------------------------------------------------------
    def set_depth_host(self, slot: int, src_depth: np.ndarray, tgt_depth: np.ndarray):
        sd = np.ascontiguousarray(src_depth, dtype=np.uint16)
        td = np.ascontiguousarray(tgt_depth, dtype=np.uint16)
        self._check(self.lib.slam3d_icp_set_depth_host(self._h, C.c_int32(slot), _vp(sd), _vp(td)), False)

    def set_clouds_host(self, slot: int, src: np.ndarray, tgt: np.ndarray):
        W, H = self.params.width, self.params.height
        s = np.ascontiguousarray(src, dtype=np.float32); t = np.ascontiguousarray(tgt, dtype=np.float32)
        sv, tv = _cloud_view(s, W, H), _cloud_view(t, W, H)
        self._check(self.lib.slam3d_icp_set_clouds_host(self._h, C.c_int32(slot), C.byref(sv), C.byref(tv)), False)

    def set_clouds_device(self, slot: int, d_src_ptr: int, d_tgt_ptr: int):
        self._check(self.lib.slam3d_icp_set_clouds_device(self._h, C.c_int32(slot), C.c_void_p(d_src_ptr), C.c_void_p(d_tgt_ptr)), False)

    def set_depth_device(self, slot: int, d_src_ptr: int, d_tgt_ptr: int):
        self._check(self.lib.slam3d_icp_set_depth_device(self._h, C.c_int32(slot), C.c_void_p(d_src_ptr), C.c_void_p(d_tgt_ptr)), False)

    # ---- resident frames -------------------------------------------------------------
    def frame_count(self) -> int:
        return int(self.lib.slam3d_icp_frame_count(self._h))

    def first_free_frame(self) -> int:
        """id of the first caller-owned frame (the 2*max_batch implicit ones come first)"""
        return 2 * self.params.max_batch

    def frame_set_depth_host(self, frame: int, depth: np.ndarray, keep: bool = True):
        d = np.ascontiguousarray(depth, dtype=np.uint16)
        if keep:        # the copy is asynchronous: the array must outlive it
            self._keep_frames = getattr(self, "_keep_frames", {})
            self._keep_frames[frame] = d
        self._check(self.lib.slam3d_icp_frame_set_depth_host(self._h, C.c_int32(frame), _vp(d)), False)

    def frame_set_depth_host_ptr(self, frame: int, ptr: int):
        """depth image at a raw host address (pinned staging owned by the caller)"""
        self._check(self.lib.slam3d_icp_frame_set_depth_host(self._h, C.c_int32(frame), C.c_void_p(ptr)), False)

    def frame_set_depth_device(self, frame: int, d_ptr: int):
        self._check(self.lib.slam3d_icp_frame_set_depth_device(self._h, C.c_int32(frame), C.c_void_p(d_ptr)), False)

    def frame_set_cloud_host(self, frame: int, cloud: np.ndarray):
        c = np.ascontiguousarray(cloud, dtype=np.float32)
        self._keep_frames = getattr(self, "_keep_frames", {})
        self._keep_frames[frame] = c
        cv = _cloud_view(c, self.params.width, self.params.height)
        self._check(self.lib.slam3d_icp_frame_set_cloud_host(self._h, C.c_int32(frame), C.byref(cv)), False)

    def frame_set_cloud_device(self, frame: int, d_ptr: int):
        self._check(self.lib.slam3d_icp_frame_set_cloud_device(self._h, C.c_int32(frame), C.c_void_p(d_ptr)), False)

    def frame_invalidate(self, frame: int):
        """the contents of a borrowed device buffer changed in place: rebuild the frame's normals / tiles at the next run"""
        self._check(self.lib.slam3d_icp_frame_invalidate(self._h, C.c_int32(frame)), False)

    def set_pair(self, slot: int, src_frame: int, tgt_frame: int):
        self._check(self.lib.slam3d_icp_set_pair(self._h, C.c_int32(slot), C.c_int32(src_frame), C.c_int32(tgt_frame)), False)

    def run(self, B: int, T_init=None, stream: int = 0):
        Ti = None if T_init is None else np.ascontiguousarray(T_init, dtype=np.float64).reshape(B, 16)
        self._check(self.lib.slam3d_icp_run(self._h, C.c_int32(B), _vp(Ti), C.c_void_p(stream)), False)

    def fetch_results(self, B: int) -> list:
        out = (Result * B)()
        self._check(self.lib.slam3d_icp_fetch_results(self._h, C.c_int32(B), out), False)
        return [r.as_dict() for r in out]

    # ---- introspection -----------------------------------------------------------------
    def get_correspondences(self, slot: int = 0):
        idx = np.empty(self.N, dtype=np.int32); d2 = np.empty(self.N, dtype=np.float32)
        self._check(self.lib.slam3d_icp_get_correspondences(self._h, C.c_int32(slot), _vp(idx), _vp(d2)), False)
        return idx, d2

    def set_corr_trace(self, on: bool = True):
        self._check(self.lib.slam3d_icp_set_corr_trace(self._h, int(bool(on))), False)

    def get_correspondences_at(self, it: int, slot: int = 0) -> np.ndarray:
        idx = np.empty(self.N, dtype=np.int32)
        self._check(self.lib.slam3d_icp_get_correspondences_at(self._h, C.c_int32(slot), C.c_int32(it), _vp(idx)), False)
        return idx

    def get_trace(self, slot: int = 0):
        it = self.params.iterations
        Tt = np.zeros((it + 1, 16), dtype=np.float64); St = np.zeros((max(it, 1), NSUMS), dtype=np.float64)
        self._check(self.lib.slam3d_icp_get_trace(self._h, C.c_int32(slot), _vp(Tt), _vp(St)), False)
        return Tt.reshape(-1, 4, 4), St[:it]

    def get_clouds(self, slot: int = 0, normals: bool = True):
        H, W = self.params.height, self.params.width
        s = np.empty((H, W, 4), dtype=np.float32); t = np.empty((H, W, 4), dtype=np.float32)
        n = np.empty((H, W, 4), dtype=np.float32) if normals else None
        self._check(self.lib.slam3d_icp_get_clouds(self._h, C.c_int32(slot), _vp(s), _vp(t), _vp(n)), False)
        return s, t, n

    def set_profiling(self, on: bool = True):
        self._check(self.lib.slam3d_icp_set_profiling(self._h, int(bool(on))), False)

    def get_timings(self):
        ms = (C.c_float * 4)()
        self._check(self.lib.slam3d_icp_get_timings(self._h, ms), False)
        return dict(preprocess_ms=ms[0], nn_ms=ms[1], accumulate_solve_ms=ms[2], total_ms=ms[3])

    def get_iteration_timings(self) -> np.ndarray:
        ms = np.zeros(max(self.params.iterations, 1), dtype=np.float32)
        self._check(self.lib.slam3d_icp_get_iteration_timings(self._h, _vp(ms)), False)
        return ms[: self.params.iterations]

    def set_stamping(self, ring_runs: int = 64):
        """launch stamps of the last `ring_runs` runs stay on the device (0: off)"""
        self._check(self.lib.slam3d_icp_set_stamping(self._h, C.c_int32(int(ring_runs))), False)
        self._stamp_ring = int(ring_runs)

    def get_stamps(self, max_runs: int = 0) -> np.ndarray:
        """(start, end) ticks (10 ns, device real-time counter) of the launches of the last runs, oldest first:
        [runs, 2 * iterations, 2] uint64, rows [0, iterations) NN launches, then the solve launches"""
        max_runs = max_runs or getattr(self, "_stamp_ring", 0) or 1
        rows = 2 * max(self.params.iterations, 1)
        out = np.zeros((max_runs, rows, 2), dtype=np.uint64)
        n = C.c_int32(0)
        self._check(self.lib.slam3d_icp_get_stamps(self._h, _vp(out), C.c_int32(max_runs), C.byref(n)), False)
        return out[: n.value]

    def get_list_debug(self) -> np.ndarray:
        """SLAM3D_LIST_DEBUG=1, point-list handles (one pair per launch): thread 0 of every block's phase times of the last run,
        [iterations][256 blocks][12] in 10 ns ticks: bounds, listing, scans, rows, Gram, arrive, barrier wait, totals, derive, solve;
        [10] tiles wave 0 scanned, [11] iteration start (blocks beyond the grid read 0)"""
        it = max(self.params.iterations, 1)
        out = np.zeros(it * 256 * 12, dtype=np.int64)
        self._check(self.lib.slam3d_icp_get_nn_debug(self._h, _vp(out), C.c_int32(out.size)), False)
        return out.reshape(it, 256, 12)

    def get_nn_debug(self) -> np.ndarray:
        nt = ((self.params.width + 7) // 8) * ((self.params.height + 7) // 8)
        out = np.zeros(nt * 20, dtype=np.int64)
        self._check(self.lib.slam3d_icp_get_nn_debug(self._h, _vp(out), C.c_int32(out.size)), False)
        return out.reshape(nt, 20)

    # ---- building blocks ---------------------------------------------------------------
    def backproject_u16(self, depth: np.ndarray) -> np.ndarray:
        d = np.ascontiguousarray(depth, dtype=np.uint16)
        out = np.empty((self.params.height, self.params.width, 4), dtype=np.float32)
        self._check(self.lib.slam3d_backproject_u16(self._h, _vp(d), _vp(out)), False)
        return out

    def fit_planes(self, cloud: np.ndarray, labels: np.ndarray, nplanes: int):
        c = np.ascontiguousarray(cloud, dtype=np.float32)
        cv = _cloud_view(c, self.params.width, self.params.height)
        lab = np.ascontiguousarray(labels, dtype=np.int32).reshape(-1)
        planes = (Plane * nplanes)()
        self._check(self.lib.slam3d_fit_planes(self._h, C.byref(cv), _vp(lab), C.c_int32(nplanes), planes), False)
        return [dict(coeff=np.array(p.coeff), count=p.count, centroid=np.array(p.centroid)) for p in planes]

    # ---- frame ingestion filters (f-1) ------------------------------------------------------
    def voxel_grid(self, pts16: np.ndarray, leaf: float = 0.03) -> np.ndarray:
        pts = np.ascontiguousarray(pts16, dtype=np.float32).reshape(-1, 4)
        out = np.zeros_like(pts)
        m = C.c_int32(0)
        self._check(self.lib.slam3d_voxel_grid(self._h, _vp(pts), C.c_int32(pts.shape[0]), C.c_float(leaf), _vp(out), C.byref(m)), False)
        return out[: m.value].copy()

    def voxel_grid_only(self, pts16: np.ndarray, leaf: float = 0.03) -> np.ndarray:
        pts = np.ascontiguousarray(pts16, dtype=np.float32).reshape(-1, 4)
        out = np.zeros_like(pts)
        m = C.c_int32(0)
        self._check(self.lib.slam3d_voxel_grid_only(self._h, _vp(pts), C.c_int32(pts.shape[0]), C.c_float(leaf), _vp(out), C.byref(m)), False)
        return out[: m.value].copy()

    def pass_transform(self, pts16: np.ndarray, T: np.ndarray, z_max: float = 5.0):
        pts = np.ascontiguousarray(pts16, dtype=np.float32).reshape(-1, 4)
        Tm = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        out = np.zeros_like(pts)
        m = C.c_int32(0)
        self._check(self.lib.slam3d_pass_transform(self._h, _vp(pts), C.c_int32(pts.shape[0]), C.c_float(z_max), _vp(Tm), _vp(out), C.byref(m)), False)
        return out, m.value

    def voxel_grid_device(self, d_pts: int, n: int, d_out: int, leaf: float = 0.03, stream: int = 0) -> int:
        m = C.c_int32(0)
        self._check(self.lib.slam3d_voxel_grid_device(self._h, C.c_void_p(d_pts), C.c_int32(n), C.c_float(leaf), C.c_void_p(d_out),
                                                      C.byref(m), C.c_void_p(stream)), False)
        return m.value

    def voxel_grid_batch_device(self, d_pts: Sequence[int], n: Sequence[int], d_out: Sequence[int], leaf: float = 0.03, stream: int = 0) -> list:
        """B clouds resident on the device (pointers), one launch sequence; -> voxel count per cloud"""
        B = len(d_pts)
        pin = (C.c_void_p * B)(*[C.c_void_p(p) for p in d_pts])
        pout = (C.c_void_p * B)(*[C.c_void_p(p) for p in d_out])
        nn = (C.c_int32 * B)(*[int(x) for x in n])
        m = (C.c_int32 * B)()
        self._check(self.lib.slam3d_voxel_grid_batch_device(self._h, C.c_int32(B), pin, nn, C.c_float(leaf), pout, m, C.c_void_p(stream)), False)
        return list(m)

    # ---- plane segmentation (f-2) ---------------------------------------------------------
    def voxel_grid_path_counts(self):
        """(calls ordered by the dense path alone, calls that also took the general ordering path) of this handle so far"""
        c = (C.c_int64 * 2)()
        self._check(self.lib.slam3d_voxel_grid_path_counts(self._h, c), False)
        return int(c[0]), int(c[1])

    @staticmethod
    def seg_params(**kw) -> SegParams:
        sp = SegParams()
        load_library().slam3d_seg_default_params(C.byref(sp))
        for k, v in kw.items():
            if not hasattr(sp, k):
                raise AttributeError(k)
            setattr(sp, k, v)
        return sp

    @staticmethod
    def _planes_out(planes, nplanes, B, maxp):
        return [[dict(coeff=np.array(planes[b * maxp + r].coeff), count=planes[b * maxp + r].count,
                      centroid=np.array(planes[b * maxp + r].centroid)) for r in range(nplanes[b])] for b in range(B)]

    def segment_planes(self, xyz4: np.ndarray, sp: Optional[SegParams] = None, want_labels: bool = True):
        sp = sp or self.seg_params()
        c = np.ascontiguousarray(xyz4, dtype=np.float32)
        cv = _cloud_view(c, self.params.width, self.params.height)
        planes = (Plane * sp.max_planes)()
        npl = (C.c_int32 * 1)()
        lab = np.zeros(self.params.width * self.params.height, dtype=np.int32) if want_labels else None
        self._check(self.lib.slam3d_segment_planes(self._h, C.byref(cv), C.byref(sp), planes, npl, _vp(lab)), False)
        return self._planes_out(planes, npl, 1, sp.max_planes)[0], lab

    def segment_planes_device(self, d_cloud_ptrs: Sequence[int], sp: Optional[SegParams] = None, d_labels: int = 0, stream: int = 0):
        sp = sp or self.seg_params()
        B = len(d_cloud_ptrs)
        ptrs = (C.c_void_p * B)(*[C.c_void_p(p) for p in d_cloud_ptrs])
        planes = (Plane * (sp.max_planes * B))()
        npl = (C.c_int32 * B)()
        self._check(self.lib.slam3d_segment_planes_device(self._h, C.c_int32(B), ptrs, C.byref(sp), planes, npl,
                                                          C.c_void_p(d_labels), C.c_void_p(stream)), False)
        return self._planes_out(planes, npl, B, sp.max_planes)

    # ---- SLAM3D_EST_PLANE ----------------------------------------------------------------
    def set_seg_params(self, sp: SegParams):
        self._check(self.lib.slam3d_icp_set_seg_params(self._h, C.byref(sp)), False)

    def get_frame_planes(self, frame: int) -> list:
        planes = (Plane * 8)()
        n = C.c_int32(0)
        self._check(self.lib.slam3d_icp_get_frame_planes(self._h, C.c_int32(frame), planes, C.byref(n)), False)
        return [dict(coeff=np.array(planes[r].coeff), count=planes[r].count, centroid=np.array(planes[r].centroid)) for r in range(n.value)]

    def get_plane_assoc(self, slot: int = 0) -> np.ndarray:
        a = np.zeros(8, dtype=np.int32)
        self._check(self.lib.slam3d_icp_get_plane_assoc(self._h, C.c_int32(slot), _vp(a)), False)
        return a

    # ---- dense mode --------------------------------------------------------------------
    def dense_set_rows(self, r0: int, r1: int):
        self._check(self.lib.slam3d_icp_dense_set_rows(self._h, C.c_int32(r0), C.c_int32(r1)), False)

    def dense_begin(self, T_init=None, stream: int = 0):
        Ti = None if T_init is None else np.ascontiguousarray(T_init, dtype=np.float64).reshape(16)
        self._check(self.lib.slam3d_icp_dense_begin(self._h, _vp(Ti), C.c_void_p(stream)), False)

    def dense_partial(self, stream: int = 0) -> np.ndarray:
        s = np.zeros(NRAW, dtype=np.int64)           # exact integer Gram totals: order-free
        self._check(self.lib.slam3d_icp_dense_partial(self._h, _vp(s), C.c_void_p(stream)), False)
        return s

    def dense_update(self, sums: np.ndarray, stream: int = 0):
        s = np.ascontiguousarray(sums, dtype=np.int64).reshape(NRAW)
        self._check(self.lib.slam3d_icp_dense_update(self._h, _vp(s), C.c_void_p(stream)), False)

    def dense_partial_device(self, d_sums: int, stream: int = 0):
        self._check(self.lib.slam3d_icp_dense_partial_device(self._h, C.c_void_p(d_sums), C.c_void_p(stream)), False)

    def dense_update_device(self, d_sums: int, stream: int = 0):
        self._check(self.lib.slam3d_icp_dense_update_device(self._h, C.c_void_p(d_sums), C.c_void_p(stream)), False)

    def dense_finish_device(self, d_sums: int, stream: int = 0) -> dict:
        out = Result()
        self._check(self.lib.slam3d_icp_dense_finish_device(self._h, C.c_void_p(d_sums), C.c_void_p(stream), C.byref(out)), False)
        return out.as_dict()

    def dense_run(self, comm: "Optional[Comm]" = None, T_init=None) -> dict:
        """BASELINE config 5 inside the library: rows sharded over comm's ranks, ncclAllReduce per iteration"""
        Ti = None if T_init is None else np.ascontiguousarray(T_init, dtype=np.float64).reshape(16)
        out = Result()
        self._check(self.lib.slam3d_icp_dense_run(self._h, comm._c if comm is not None else None, _vp(Ti), C.byref(out)))
        return out.as_dict()

    def set_fault_injection(self, dense_fail_at: int = -1) -> None:
        """test hook: this handle's dense runs fail in iteration `dense_fail_at` (-1: off)"""
        self._check(self.lib.slam3d_icp_set_fault_injection(self._h, C.c_int32(dense_fail_at)))

    def dense_run_with(self, rank: int, world: int, allreduce, T_init=None) -> dict:
        """slam3d_icp_dense_run over a caller's transport: allreduce(d_buf: int, count: int, stream: int) -> 0 must SUM `count`
        int64 at device address d_buf over the ranks, in place (dense.host_staged_allreduce: through torch.distributed / gloo)"""
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

        def thunk(ctx, d_buf, count, stream):
            try:
                return int(allreduce(int(d_buf or 0), int(count), int(stream or 0)))
            except Exception:       # noqa: BLE001 -- an exception must not unwind through the C frames
                return 1
        cb = CB(thunk) if allreduce is not None else C.cast(None, CB)
        Ti = None if T_init is None else np.ascontiguousarray(T_init, dtype=np.float64).reshape(16)
        out = Result()
        self._check(self.lib.slam3d_icp_dense_run_with(self._h, C.c_int32(rank), C.c_int32(world), cb, None, _vp(Ti), C.byref(out)))
        return out.as_dict()

    def dense_finish(self, last_sums: np.ndarray) -> dict:
        s = np.ascontiguousarray(last_sums, dtype=np.int64).reshape(NRAW)
        out = Result()
        self._check(self.lib.slam3d_icp_dense_finish(self._h, _vp(s), C.byref(out)), False)
        return out.as_dict()


def shard_range(n: int, world: int, rank: int):
    b, e = C.c_int32(0), C.c_int32(0)
    load_library().slam3d_shard_range(C.c_int32(n), C.c_int32(world), C.c_int32(rank), C.byref(b), C.byref(e))
    return b.value, e.value


def comm_unique_id() -> bytes:
    """rank 0: the 128-byte id every rank passes to Comm(...)"""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    rc = load_library().slam3d_comm_get_unique_id(buf)
    if rc:
        raise Slam3dError(rc, "slam3d_comm_get_unique_id: " + load_library().slam3d_comm_last_error(None).decode())
    return bytes(buf)


class Comm:
    """RAII wrapper of slam3d_comm (RCCL communicator behind the C-ABI)."""

    def __init__(self, uid: bytes, rank: int, world: int, device: int):
        self.lib = load_library()
        assert len(uid) == COMM_ID_BYTES
        self._c = C.c_void_p()
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(uid)
        rc = self.lib.slam3d_comm_init(buf, C.c_int32(rank), C.c_int32(world), C.c_int32(device), C.byref(self._c))
        if rc:
            raise Slam3dError(rc, "slam3d_comm_init: " + self.lib.slam3d_comm_last_error(None).decode())
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self.lib.slam3d_comm_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise Slam3dError(rc, self.lib.slam3d_strerror(rc).decode() + ": " + self.lib.slam3d_comm_last_error(self._c).decode())

    @staticmethod
    def records(results) -> "C.Array":
        rec = (PoseRecord * len(results))()
        for i, r in enumerate(results):
            rec[i].T[:] = np.asarray(r["T"], dtype=np.float64).reshape(16).tolist()
            rec[i].norm = r["norm"]; rec[i].inliers = r["inliers"]; rec[i].status = r["status"]; rec[i].rmse = r.get("rmse", 0.0)
        return rec

    def gather_submit(self, results):
        rec = self.records(results)
        self._n = len(results)
        self._check(self.lib.slam3d_pose_gather_submit(self._c, rec, C.c_int32(len(results))))

    def gather_collect(self, n_local: Optional[int] = None):
        n = n_local if n_local is not None else self._n
        out = (PoseRecord * (n * self.world))()
        self._check(self.lib.slam3d_pose_gather_collect(self._c, out))
        return [dict(T=np.array(r.T).reshape(4, 4), norm=r.norm, inliers=r.inliers, status=r.status, rmse=r.rmse) for r in out]

    def gather(self, results):
        self.gather_submit(results)
        return self.gather_collect(len(results))

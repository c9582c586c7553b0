"""ctypes binding of oracle/liboracle.so -- the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
NSUMS = 29


class OrcParams(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("depth_factor", C.c_double), ("z_filter", C.c_double),
        ("iterations", C.c_int), ("max_corr_dist", C.c_double), ("estimator", C.c_int),
        ("normal_window", C.c_int), ("normal_min_inliers", C.c_int), ("normal_inlier_dist", C.c_double),
        ("min_inliers", C.c_int), ("error_threshold", C.c_double),
        ("nn_method", C.c_int), ("threads", C.c_int),
        ("max_plane_residual2", C.c_double), ("min_normal_cos", C.c_double),
        ("coarse_iterations", C.c_int),
        ("seg_distance_threshold", C.c_float), ("seg_plane_percent", C.c_float),
        ("seg_max_planes", C.c_int), ("seg_hypotheses", C.c_int), ("seg_seed", C.c_uint64),
        ("plane_pair_gate", C.c_int), ("plane_only", C.c_int),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16), ("norm", C.c_double), ("rmse", C.c_double),
        ("inliers", C.c_int), ("status", C.c_int), ("iterations", C.c_int),
        ("n_src", C.c_int), ("n_tgt", C.c_int),
    ]


_lib = None


def build(force: bool = False) -> str:
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "icp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_icp.restype = C.c_int
        _lib.orc_nn_once.restype = C.c_int
        _lib.orc_nn_once_ex.restype = C.c_int
        _lib.orc_solve6.restype = C.c_int
    return _lib


def _fp(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def params(intr=None, **kw) -> OrcParams:
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    if intr is not None:
        p.width, p.height = intr.width, intr.height
        p.fx, p.fy, p.cx, p.cy, p.depth_factor = intr.fx, intr.fy, intr.cx, intr.cy, intr.depth_factor
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def backproject(depth: np.ndarray, p: OrcParams) -> np.ndarray:
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    out = np.empty((p.height, p.width, 4), dtype=np.float32)
    lib().orc_backproject(_fp(depth, C.c_uint16), C.byref(p), _fp(out, C.c_float))
    return out


def normals(xyz4: np.ndarray, p: OrcParams) -> np.ndarray:
    xyz4 = np.ascontiguousarray(xyz4, dtype=np.float32)
    out = np.empty((p.height, p.width, 4), dtype=np.float32)
    lib().orc_normals(_fp(xyz4, C.c_float), C.byref(p), _fp(out, C.c_float))
    return out


def plane_normals(xyz4: np.ndarray, p: OrcParams):
    """spec S2p: -> (nrm4 [H,W,4] with w = 1 + plane, planes [n,8] = a b c d cx cy cz count, labels [N])"""
    xyz4 = np.ascontiguousarray(xyz4, dtype=np.float32)
    out = np.empty((p.height, p.width, 4), dtype=np.float32)
    planes = np.zeros((p.seg_max_planes, 8), dtype=np.float32)
    labels = np.zeros(p.width * p.height, dtype=np.int32)
    f = lib().orc_plane_normals
    f.restype = C.c_int
    k = f(_fp(xyz4, C.c_float), C.byref(p), _fp(out, C.c_float), _fp(planes, C.c_float), _fp(labels, C.c_int32))
    return out, planes[:k].copy(), labels


def plane_assoc(planes1, planes2, T=None) -> np.ndarray:
    a = np.ascontiguousarray(planes1, dtype=np.float32).reshape(-1, 8)
    b = np.ascontiguousarray(planes2, dtype=np.float32).reshape(-1, 8)
    out = np.full(max(1, a.shape[0]), -1, dtype=np.int32)
    Tm = np.ascontiguousarray(T, dtype=np.float64).reshape(16) if T is not None else None
    f = lib().orc_plane_assoc
    f.restype = None
    f(_fp(a, C.c_float), C.c_int(a.shape[0]), _fp(b, C.c_float), C.c_int(b.shape[0]), _fp(Tm, C.c_double), _fp(out, C.c_int32))
    return out[: a.shape[0]]


def icp(src4: np.ndarray, tgt4: np.ndarray, p: OrcParams, T_init=None, trace: bool = True):
    """Returns dict(T, norm, rmse, inliers, status, n_src, n_tgt, idx, d2, T_trace, sums_trace)."""
    N = p.width * p.height
    src4 = np.ascontiguousarray(src4, dtype=np.float32)
    tgt4 = np.ascontiguousarray(tgt4, dtype=np.float32)
    res = OrcResult()
    idx = np.empty(N, dtype=np.int32)
    d2 = np.empty(N, dtype=np.float32)
    Ttr = np.zeros((p.iterations + 1, 16), dtype=np.float64) if trace else None
    Str = np.zeros((max(p.iterations, 1), NSUMS), dtype=np.float64) if trace else None
    Ti = np.ascontiguousarray(T_init, dtype=np.float64).reshape(16) if T_init is not None else None
    lib().orc_icp(_fp(src4, C.c_float), _fp(tgt4, C.c_float), C.byref(p), _fp(Ti, C.c_double), C.byref(res),
                  _fp(idx, C.c_int32), _fp(d2, C.c_float), _fp(Ttr, C.c_double), _fp(Str, C.c_double))
    return dict(T=np.array(res.T).reshape(4, 4), norm=res.norm, rmse=res.rmse, inliers=res.inliers,
                status=res.status, n_src=res.n_src, n_tgt=res.n_tgt, idx=idx, d2=d2,
                T_trace=Ttr.reshape(-1, 4, 4) if trace else None, sums_trace=Str[:p.iterations] if trace else None)


def nn_once(src4, tgt4, p: OrcParams, T=None, use_normals=False, coarse: bool = False):
    """use_normals: False / True (7x7-window normals) / 2 (per-plane normals, spec S2p)"""
    N = p.width * p.height
    src4 = np.ascontiguousarray(src4, dtype=np.float32)
    tgt4 = np.ascontiguousarray(tgt4, dtype=np.float32)
    idx = np.empty(N, dtype=np.int32)
    d2 = np.empty(N, dtype=np.float32)
    Ti = np.ascontiguousarray(T, dtype=np.float64).reshape(16) if T is not None else None
    ns = lib().orc_nn_once_ex(_fp(src4, C.c_float), _fp(tgt4, C.c_float), C.byref(p), _fp(Ti, C.c_double),
                              C.c_int(int(use_normals)), C.c_int(1 if coarse else 0), _fp(idx, C.c_int32), _fp(d2, C.c_float))
    return idx, d2, ns


def fit_planes(xyz4, labels, nplanes: int):
    xyz4 = np.ascontiguousarray(xyz4, dtype=np.float32).reshape(-1, 4)
    labels = np.ascontiguousarray(labels, dtype=np.int32).reshape(-1)
    planes = np.zeros((nplanes, 4), dtype=np.float32)
    counts = np.zeros(nplanes, dtype=np.int32)
    lib().orc_fit_planes(_fp(xyz4, C.c_float), _fp(labels, C.c_int32), C.c_int(labels.size), C.c_int(nplanes),
                         _fp(planes, C.c_float), _fp(counts, C.c_int32))
    return planes, counts


class SegParams(C.Structure):
    _fields_ = [("distance_threshold", C.c_float), ("plane_percent", C.c_float), ("max_planes", C.c_int32),
                ("hypotheses", C.c_int32), ("seed", C.c_uint64)]


def segment_planes(xyz4, zmax=7.0, distance_threshold=0.08, plane_percent=0.2, max_planes=3, hypotheses=64, seed=1):
    """-> (planes[nplanes] dicts(coeff, centroid, count), labels[N])"""
    xyz4 = np.ascontiguousarray(xyz4, dtype=np.float32).reshape(-1, 4)
    n = xyz4.shape[0]
    sp = SegParams(distance_threshold, plane_percent, max_planes, hypotheses, seed)
    planes = np.zeros((max_planes, 8), dtype=np.float32)
    labels = np.zeros(n, dtype=np.int32)
    f = lib().orc_segment_planes
    f.restype = C.c_int
    k = f(_fp(xyz4, C.c_float), C.c_int(n), C.c_float(zmax), C.byref(sp), _fp(planes, C.c_float), _fp(labels, C.c_int32))
    return [dict(coeff=planes[i, :4].copy(), centroid=planes[i, 4:7].copy(), count=int(planes[i, 7])) for i in range(k)], labels


def voxel_grid(pts16, leaf=0.03, zmax=7.0):
    """pts16: (n,4) float32 records {x,y,z,rgba bits} -> (m,4) float32 records, ascending (iz,iy,ix)"""
    pts = np.ascontiguousarray(pts16, dtype=np.float32).reshape(-1, 4)
    out = np.zeros_like(pts)
    f = lib().orc_voxel_grid
    f.restype = C.c_int
    m = f(_fp(pts, C.c_float), C.c_int(pts.shape[0]), C.c_float(leaf), C.c_float(zmax), _fp(out, C.c_float))
    return out[:m].copy()


def voxel_grid_only(pts16, leaf=0.03):
    pts = np.ascontiguousarray(pts16, dtype=np.float32).reshape(-1, 4)
    out = np.zeros_like(pts)
    f = lib().orc_voxel_grid_range
    f.restype = C.c_int
    m = f(_fp(pts, C.c_float), C.c_int(pts.shape[0]), C.c_float(leaf), C.c_float(-np.inf), C.c_float(np.inf), _fp(out, C.c_float))
    return out[:m].copy()


def pass_transform(pts16, T, z_max=5.0):
    pts = np.ascontiguousarray(pts16, dtype=np.float32).reshape(-1, 4)
    Tm = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    out = np.zeros_like(pts)
    f = lib().orc_pass_transform
    f.restype = C.c_int
    k = f(_fp(pts, C.c_float), C.c_int(pts.shape[0]), C.c_float(z_max), _fp(Tm, C.c_double), _fp(out, C.c_float))
    return out, k


def pose_error(Tref, T):
    Tref = np.ascontiguousarray(Tref, dtype=np.float64).reshape(16)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    r, t = C.c_double(), C.c_double()
    lib().orc_pose_error(_fp(Tref, C.c_double), _fp(T, C.c_double), C.byref(r), C.byref(t))
    return r.value, t.value


def eig3(A6):
    A6 = np.ascontiguousarray(A6, dtype=np.float64)
    ev = np.zeros(3); V = np.zeros(9)
    lib().orc_eig3(_fp(A6, C.c_double), _fp(ev, C.c_double), _fp(V, C.c_double))
    return ev, V.reshape(3, 3)


def smallest_evec3(A6):
    """spec S2's direction of least variance (adjugate power iteration): (ok, unit vector)"""
    A6 = np.ascontiguousarray(A6, dtype=np.float64)
    n = np.zeros(3)
    ok = lib().orc_smallest_evec3(_fp(A6, C.c_double), _fp(n, C.c_double))
    return bool(ok), n


def solve6(U21, b6):
    U21 = np.ascontiguousarray(U21, dtype=np.float64); b6 = np.ascontiguousarray(b6, dtype=np.float64)
    x = np.zeros(6)
    rc = lib().orc_solve6(_fp(U21, C.c_double), _fp(b6, C.c_double), _fp(x, C.c_double))
    return rc, x


def svd3_rotation(H9):
    H9 = np.ascontiguousarray(H9, dtype=np.float64).reshape(9)
    R = np.zeros(9)
    lib().orc_svd3_rotation(_fp(H9, C.c_double), _fp(R, C.c_double))
    return R.reshape(3, 3)


def sincos(x: float):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(C.c_double(x), C.byref(s), C.byref(c))
    return s.value, c.value

"""Deterministic synthetic Kinect-like frame pairs (BASELINE.md section 4 workloads).

Scene (camera-1 frame == world, x right, y down, z forward -- the convention of the
reference's back-projection, src/convert2PCD.cpp:65-69): floor, back wall with a doorway
(rays through it end beyond ``z_filter`` => invalid, like the far range of the reference
fixtures), left side wall, four axis-aligned boxes on the floor.  Frame 2 sees the same
scene from ``T_gt`` (X_frame2 = T_gt . X_frame1, the direction of ``multiPnP``'s result,
src/GraphicEnd.cpp:557-659).

Everything is computed from +,-,*,/,sqrt,floor and integer hashing only (no libm
transcendental), so the u16 depth images are bit-identical on every host; the golden
fixtures under tests/golden store their SHA-256.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser, vectorised over uint64 counters."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n doubles in [0,1): counter-based, independent per (seed, stream)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream)], dtype=np.uint64))[0]
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    return (_splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _gauss(seed: int, stream: int, n: int) -> np.ndarray:
    """~N(0,1): Irwin-Hall sum of 12 uniforms - 6 (basic ops only)."""
    u = _uniform(seed, stream, 12 * n).reshape(12, n)
    acc = u[0].copy()
    for k in range(1, 12):          # explicit order: no pairwise/BLAS reassociation
        acc += u[k]
    return acc - 6.0


@dataclass
class Intrinsics:
    width: int = 640
    height: int = 480
    fx: float = 525.0      # src/convert2PCD.cpp:19-23
    fy: float = 525.0
    cx: float = 319.5
    cy: float = 235.5
    depth_factor: float = 1000.0

    @staticmethod
    def scaled(width: int, height: int) -> "Intrinsics":
        """Kinect intrinsics rescaled to another resolution with the same field of view."""
        s = width / 640.0
        return Intrinsics(width, height, 525.0 * s, 525.0 * s, (319.5 + 0.5) * s - 0.5, (235.5 + 0.5) * s - 0.5, 1000.0)


def pose_from_seed(seed: int, max_angle_deg: float = 3.0, max_trans: float = 0.05) -> np.ndarray:
    """Random small SE(3): rotation from a normalised quaternion (1, v), |v| <= tan(max/2)."""
    u = _uniform(seed, 7, 8)
    v = 2.0 * u[0:3] - 1.0
    nv = np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    if nv < 1e-9:
        v = np.array([1.0, 0.0, 0.0]); nv = 1.0
    # tan(1.5 deg) = 0.0261859...; scale linearly in tan(half-angle)
    half_tan_max = 0.026185921569186924 * (max_angle_deg / 3.0)
    v = v / nv * (u[3] * half_tan_max)
    q = np.array([1.0, v[0], v[1], v[2]])
    q = q / np.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    w, x, y, z = q
    R = np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])
    t = (2.0 * u[4:7] - 1.0) * max_trans
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


# scene description -------------------------------------------------------------------
_FLOOR_Y = 1.2
_BACK_Z = 4.5
_FAR_Z = 8.5            # seen through the doorway; > z_filter (7.0) => invalid
_LEFT_X = -2.0
_DOOR = (0.8, 1.6, -0.8, 1.2)   # x0, x1, y0, y1 opening in the back wall
_BOXES = [  # (xmin, xmax, ymin, ymax, zmin, zmax) resting on the floor
    (-1.2, -0.5, 0.5, 1.2, 2.2, 2.9),
    (0.3, 0.9, 0.7, 1.2, 1.8, 2.3),
    (-0.3, 0.25, 0.2, 1.2, 3.3, 3.8),
    (1.2, 2.0, 0.6, 1.2, 3.0, 3.9),
]


def _raycast(o: np.ndarray, d: np.ndarray, want_ids: bool = False):
    """o (3,), d (n,3) world rays -> smallest positive hit parameter t (n,)
    (and, with want_ids, the surface id: 0 floor, 1 left wall, 2 back wall, -1 anything else)."""
    n = d.shape[0]
    t_best = np.full(n, np.inf)
    ids = np.full(n, -1, dtype=np.int32)
    with np.errstate(divide="ignore", invalid="ignore"):
        # floor y = FLOOR_Y
        t = (_FLOOR_Y - o[1]) / d[:, 1]
        ok = (d[:, 1] > 0) & (t > 0)
        ids = np.where(ok & (t < t_best), 0, ids)
        t_best = np.where(ok & (t < t_best), t, t_best)
        # left wall x = LEFT_X
        t = (_LEFT_X - o[0]) / d[:, 0]
        ok = (d[:, 0] < 0) & (t > 0)
        ids = np.where(ok & (t < t_best), 1, ids)
        t_best = np.where(ok & (t < t_best), t, t_best)
        # back wall z = BACK_Z with doorway
        t = (_BACK_Z - o[2]) / d[:, 2]
        hx = o[0] + t * d[:, 0]
        hy = o[1] + t * d[:, 1]
        door = (hx > _DOOR[0]) & (hx < _DOOR[1]) & (hy > _DOOR[2]) & (hy < _DOOR[3])
        ok = (d[:, 2] > 0) & (t > 0) & ~door
        ids = np.where(ok & (t < t_best), 2, ids)
        t_best = np.where(ok & (t < t_best), t, t_best)
        # far wall behind the doorway
        t = (_FAR_Z - o[2]) / d[:, 2]
        ok = (d[:, 2] > 0) & (t > 0)
        ids = np.where(ok & (t < t_best), -1, ids)
        t_best = np.where(ok & (t < t_best), t, t_best)
        # boxes (slab test)
        for b in _BOXES:
            lo = np.array([b[0], b[2], b[4]])
            hi = np.array([b[1], b[3], b[5]])
            t0 = (lo - o) / d
            t1 = (hi - o) / d
            tn = np.minimum(t0, t1).max(axis=1)
            tf = np.maximum(t0, t1).min(axis=1)
            ok = (tn <= tf) & (tn > 0)
            ids = np.where(ok & (tn < t_best), -1, ids)
            t_best = np.where(ok & (tn < t_best), tn, t_best)
    if want_ids:
        return t_best, ids
    return t_best


def render_depth(T_cam_from_world: np.ndarray, intr: Intrinsics, seed: int, stream: int,
                 noise: bool = True, holes: bool = True, z_min: float = 0.7, z_max: float = 7.0,
                 hole_block: int = 32, hole_prob: float = 0.2, noise_sigma: float = 0.0002,
                 want_labels: bool = False):
    """Depth image (uint16, millimetres for depth_factor 1000) seen by a camera whose pose
    maps world -> camera coordinates as X_cam = T . X_world."""
    W, H = intr.width, intr.height
    R = T_cam_from_world[:3, :3]
    t = T_cam_from_world[:3, 3]
    # camera centre in world, -R^T t, written out (no BLAS: bit-identical on every host)
    o = np.array([-(R[0, k] * t[0] + R[1, k] * t[1] + R[2, k] * t[2]) for k in range(3)])
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    dc = np.stack([(uu - intr.cx) / intr.fx, (vv - intr.cy) / intr.fy, np.ones_like(uu)], axis=-1).reshape(-1, 3)
    dw = np.stack([dc[:, 0] * R[0, k] + dc[:, 1] * R[1, k] + dc[:, 2] * R[2, k] for k in range(3)], axis=-1)  # R^T . dc
    z, ids = _raycast(o, dw, want_ids=True)   # camera-frame z because dc_z == 1
    if noise:
        z = z + noise_sigma * z * z * _gauss(seed, stream * 4 + 1, W * H)
    bad = ~np.isfinite(z) | (z <= z_min) | (z > z_max)
    if holes:
        bw, bh = (W + hole_block - 1) // hole_block, (H + hole_block - 1) // hole_block
        drop = (_uniform(seed, stream * 4 + 2, bw * bh) < hole_prob).reshape(bh, bw)
        drop = np.repeat(np.repeat(drop, hole_block, axis=0), hole_block, axis=1)[:H, :W].reshape(-1)
        bad |= drop
    d = np.floor(np.where(bad, 0.0, z) * intr.depth_factor + 0.5)
    d = np.where(bad, 0.0, np.clip(d, 0.0, 65535.0))
    d = d.astype(np.uint16).reshape(H, W)
    if want_labels:   # plane labels of the valid pixels (floor / left wall / back wall), -1 elsewhere
        return d, np.where(bad, -1, ids).astype(np.int32).reshape(H, W)
    return d


@dataclass
class FramePair:
    seed: int
    intr: Intrinsics
    depth_src: np.ndarray     # frame 1 = keyframe = ICP source   (H, W) uint16
    depth_tgt: np.ndarray     # frame 2 = present  = ICP target   (H, W) uint16
    T_gt: np.ndarray          # X_frame2 = T_gt . X_frame1

    def sha256(self) -> str:
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(self.depth_src).tobytes())
        h.update(np.ascontiguousarray(self.depth_tgt).tobytes())
        return h.hexdigest()


def make_pair(seed: int, width: int = 640, height: int = 480, noise: bool = True, holes: bool = True,
              max_angle_deg: float = 3.0, max_trans: float = 0.05, noise_sigma: float = 0.0002,
              hole_block: int = 32, hole_prob: float = 0.2) -> FramePair:
    """Pair ``seed`` of the BASELINE workloads (C2 = seed 1000, C3 = 1000..1063, C5 = seed 2000 @1280x960).
    ``hole_block`` (pixels at 640x480, scaled with the width) / ``hole_prob``: the invalid-pixel mask; SURVEY.md 8(d)
    specifies 8 / 0.25 (the `survey_mask` workload), the default 32 / 0.2 is DESIGN.md section 3's deviation (ii)."""
    intr = Intrinsics.scaled(width, height)
    T_gt = pose_from_seed(seed, max_angle_deg, max_trans)
    hb = max(2, int(round(hole_block * width / 640.0)))
    d1 = render_depth(np.eye(4), intr, seed, 1, noise, holes, hole_block=hb, hole_prob=hole_prob, noise_sigma=noise_sigma)
    d2 = render_depth(T_gt, intr, seed, 2, noise, holes, hole_block=hb, hole_prob=hole_prob, noise_sigma=noise_sigma)
    return FramePair(seed, intr, d1, d2, T_gt)


def backproject_numpy(depth: np.ndarray, intr: Intrinsics, z_filter: float = 7.0) -> np.ndarray:
    """Host-side S1 (same arithmetic as the spec: double, rounded once to float).  Produces the
    organized float4 cloud that is the *input* of the hot path when it is fed clouds."""
    H, W = depth.shape
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    z = depth.astype(np.float64) / intr.depth_factor
    x = (uu - intr.cx) * z / intr.fx
    y = (vv - intr.cy) * z / intr.fy
    bad = (depth == 0) | ~(z <= z_filter)
    out = np.empty((H, W, 4), dtype=np.float32)
    out[..., 0] = x; out[..., 1] = y; out[..., 2] = z; out[..., 3] = 1.0
    out[bad] = (np.nan, np.nan, np.nan, 0.0)
    return out


def _make_pair_spec(spec):
    """spec = (seed, width, height, noise_sigma[, hole_block, hole_prob])"""
    seed, width, height, sigma = spec[:4]
    kw = dict(hole_block=spec[4], hole_prob=spec[5]) if len(spec) >= 6 else {}
    return make_pair(seed, width, height, noise_sigma=sigma, **kw)


def make_pairs(specs, workers: int = 0):
    """[(seed, width, height, noise_sigma), ...] -> {spec: FramePair}, rendered by parallel worker PROCESSES started
    with subprocess (`python -m slam3d_gx_amd.synth render ...`: no fork of a process that may hold a HIP context, no
    re-import of the caller's __main__).  The ray caster is pure numpy, ~1 s per 640x480 pair on one core; results
    are bit-identical to make_pair."""
    import os
    import subprocess
    import sys
    import tempfile
    specs = list(dict.fromkeys(specs))
    workers = workers or min(len(specs), max(1, (os.cpu_count() or 2) // 2), 48)
    if workers <= 1 or len(specs) <= 1:
        return {sp: _make_pair_spec(sp) for sp in specs}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for w in range(workers):
            chunk = specs[w::workers]
            if not chunk:
                continue
            path = os.path.join(tmp, f"part{w}.npz")
            args = [sys.executable, "-m", "slam3d_gx_amd.synth", "render", path] + [",".join(repr(x) for x in sp) for sp in chunk]
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            procs.append((subprocess.Popen(args, cwd=root, env=env), path, chunk))
        for pr, path, chunk in procs:
            if pr.wait(timeout=1800) != 0:
                raise RuntimeError("synthetic frame renderer failed")
            with np.load(path) as z:
                for k, sp in enumerate(chunk):
                    out[sp] = FramePair(sp[0], Intrinsics.scaled(sp[1], sp[2]), z[f"s{k}"], z[f"t{k}"], z[f"T{k}"])
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) >= 4 and sys.argv[1] == "render":
        arrs = {}
        for k, a in enumerate(sys.argv[3:]):
            f = a.split(",")
            pr = _make_pair_spec((int(f[0]), int(f[1]), int(f[2]), float(f[3])) + ((int(f[4]), float(f[5])) if len(f) >= 6 else ()))
            arrs[f"s{k}"], arrs[f"t{k}"], arrs[f"T{k}"] = pr.depth_src, pr.depth_tgt, pr.T_gt
        np.savez(sys.argv[2], **arrs)

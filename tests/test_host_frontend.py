"""Host-side C++ mirror of the reference front end (slam3d_gx_amd/host): readers on CPU, run_SLAM on the GPU."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from slam3d_gx_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "slam3d_gx_amd", "host")

PARAMS = """%YAML:1.0
# same keys as the reference's parameters.yaml
data_source: {src}
detector_name: SIFT
descriptor_name: SIFT
start_index: 1
end_index: 100
step_time: 10
max_pos_change: {mpc}
grid_leaf: 0.03
error_threshold: 1.0
distance_threshold: 0.08
plane_percent: 0.2
max_planes: 3
loop_closure_detection: {lc}
loopclosure_frames: 4
loop_closure_error: 1.5
loop_closure_inliers: 30
lost_frames: 10
z_filter: 7.0
camera_fx: {fx}
camera_fy: {fy}
camera_cx: {cx}
camera_cy: {cy}
camera_factor: 1000.0
image_width: {w}
image_height: {h}
icp_iterations: 15
icp_extract_planes: {planes}
icp_read_pcd: {pcd}
{extra}
"""


def _build_host():
    from slam3d_gx_amd import build
    build.build_lib()
    subprocess.check_call(["make", "-C", HOST, "-s"])


def _write_png16(path, a):
    from PIL import Image
    Image.fromarray(a.astype(np.uint16)).save(path)


def test_png_and_parameter_readers(tmp_path):
    _build_host()
    pr = synth.make_pair(1000, 320, 240)
    png = tmp_path / "d.png"
    _write_png16(str(png), pr.depth_src)
    yml = tmp_path / "parameters.yaml"
    yml.write_text(PARAMS.format(src="/data/x", mpc=0.25, fx=517.0, fy=517.0, cx=318.6, cy=255.3, w=320, h=240, lc="no", planes="no", pcd="no", extra=""))
    out = subprocess.run([os.path.join(HOST, "host_selftest"), str(png), str(yml)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    flat = pr.depth_src.reshape(-1).astype(np.uint64)
    want = int((flat * (np.arange(flat.size, dtype=np.uint64) % np.uint64(9973) + np.uint64(1))).sum())
    assert f"png 320 240 {want} {int((flat != 0).sum())}" in lines
    # GetPara returns strings; unknown key -> "unknown_para_name" + stderr (src/ParameterReader.cpp:121-122)
    assert "data_source=/data/x z_filter=7.0 max_planes=3 missing=unknown_para_name fx=517.000 factor=1000.0 iters=15" in lines[-1]
    assert "Unknown parameter: no_such_key" in out.stderr


def _pcd_checksum(rec):
    b = np.ascontiguousarray(rec, dtype=np.float32).view(np.uint32).reshape(-1, 4).astype(np.uint64)
    mixed = (b[:, 0] ^ ((b[:, 1] * 3) & 0xffffffff) ^ ((b[:, 2] * 5) & 0xffffffff) ^ ((b[:, 3] * 7) & 0xffffffff))
    w = (np.arange(b.shape[0], dtype=np.uint64) % np.uint64(9973)) + np.uint64(1)
    return int((mixed * w).sum(dtype=np.uint64))


def test_pcd_reader_writer(tmp_path):
    """Row f-1: PCD v0.7 binary (the reference's format, src/convert2PCD.cpp:75-79) and ascii, incl. the reference's
    own data/exp1/pcd/1.pcd when it is present (221,202 records, 3,911 trailing bytes ignored)."""
    _build_host()
    exe = os.path.join(HOST, "host_selftest")
    rng = np.random.default_rng(4)
    rec = rng.normal(size=(1000, 4)).astype(np.float32)
    rec[:, 3] = rng.integers(0, 2 ** 32, 1000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
            "WIDTH 1000\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 1000\nDATA binary\n")
    src = tmp_path / "a.pcd"
    src.write_bytes(head.encode() + rec.tobytes() + b"trailing-garbage")
    out = subprocess.run([exe, "pcd", str(src), str(tmp_path / "b.pcd")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["pcd", "1000", "1000", "1", str(_pcd_checksum(rec))]
    assert (tmp_path / "b.pcd").read_bytes() == head.encode() + rec.tobytes()          # byte-identical re-write
    # ascii with an extra field in front and rgb as the last column
    asc = "VERSION .7\nFIELDS intensity x y z rgb\nSIZE 4 4 4 4 4\nTYPE F F F F U\nCOUNT 1 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n" \
          "9 1.5 -2 3.25 255\n8 0.5 0.25 4 16711680\n"
    (tmp_path / "c.pcd").write_text(asc)
    out = subprocess.run([exe, "pcd", str(tmp_path / "c.pcd"), str(tmp_path / "d.pcd")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    want = np.array([[1.5, -2, 3.25, 0], [0.5, 0.25, 4, 0]], dtype=np.float32)
    want[:, 3] = np.array([255, 16711680], dtype=np.uint32).view(np.float32)
    assert out.stdout.split() == ["pcd", "2", "2", "1", str(_pcd_checksum(want))]
    # failure modes: missing file, not a PCD
    assert subprocess.run([exe, "pcd", str(tmp_path / "nope.pcd"), str(tmp_path / "x.pcd")], capture_output=True).returncode == 1
    (tmp_path / "e.pcd").write_text("hello\n")
    assert subprocess.run([exe, "pcd", str(tmp_path / "e.pcd"), str(tmp_path / "x.pcd")], capture_output=True).returncode == 1
    ref = "/root/reference/data/exp1/pcd/1.pcd"
    if os.path.exists(ref):
        raw = open(ref, "rb").read()
        k = raw.index(b"DATA binary\n") + 12
        body = np.frombuffer(raw, dtype=np.float32, count=4 * 221202, offset=k).reshape(-1, 4)
        out = subprocess.run([exe, "pcd", ref, str(tmp_path / "r.pcd")], capture_output=True, text=True)
        assert out.stdout.split() == ["pcd", "221202", "221202", "1", str(_pcd_checksum(body))]
        assert (tmp_path / "r.pcd").read_bytes() == raw[: k + 16 * 221202]                # same header, same records


def test_generate_trajectory_tool(tmp_path):
    """Row f-3: generateTrajectory keyframe.txt final.g2o (src/generateTrajectory.cpp): g2o text + associate.txt time
    stamps -> TUM lines.  Pure host I/O, runs without a GPU."""
    _build_host()
    (tmp_path / "ds").mkdir()
    (tmp_path / "ds" / "associate.txt").write_text("".join(f"{1305031102.1 + 0.03 * k:.6f} rgb/{k}.png {1305031102.2 + 0.03 * k:.6f} depth/{k}.png\n" for k in range(6)))
    (tmp_path / "parameters.yaml").write_text(PARAMS.format(src=str(tmp_path / "ds"), mpc=0.25, fx=525.0, fy=525.0, cx=319.5, cy=239.5,
                                                            w=640, h=480, lc="no", planes="no", pcd="no", extra=""))
    h = np.sqrt(0.5)
    (tmp_path / "final.g2o").write_text(
        "VERTEX_SE3:QUAT 0 0 0 0 0 0 0 1\nVERTEX_SE3:QUAT 1 0.5 -0.25 1.5 0 0 %.17g %.17g\nFIX 0\n"
        "EDGE_SE3:QUAT 0 1 0.5 -0.25 1.5 0 0 %.17g %.17g 100 0 0 0 0 0 100 0 0 0 0 100 0 0 0 100 0 0 100 0 100\n" % (h, h, h, h))
    (tmp_path / "keyframe.txt").write_text("0 1\n1 4\n7 5\n")                 # vertex 7 does not exist: skipped (:64-65)
    out = subprocess.run([os.path.join(HOST, "generateTrajectory"), "keyframe.txt", "final.g2o"], cwd=str(tmp_path), capture_output=True, text=True)
    assert out.returncode == 0 and "trajectory saved." in out.stdout
    rows = [ln.split() for ln in (tmp_path / "trajectory.txt").read_text().splitlines()]
    assert len(rows) == 2 and rows[0][0] == "1305031102.100000" and rows[1][0] == "1305031102.190000"    # frames 1 and 4
    assert np.allclose([float(x) for x in rows[0][1:]], [0, 0, 0, 0, 0, 0, 1])
    assert np.allclose([float(x) for x in rows[1][1:]], [0.5, -0.25, 1.5, 0, 0, h, h], atol=1e-6)   # 90 degrees about z survives the round trip


@pytest.mark.gpu
def test_run_slam_driver_tracks_synthetic_sequence(gpu_lib, tmp_path):
    """run_SLAM N on a synthetic dep_index/ sequence: every frame is aligned against the keyframe, the robot
    pose follows the ground truth, error_of_transform.log has one norm per frame (src/GraphicEnd.cpp:232,243)."""
    _build_host()
    W, H = 320, 240
    intr = synth.Intrinsics.scaled(W, H)
    step = synth.pose_from_seed(4242, max_angle_deg=1.0, max_trans=0.02)
    data = tmp_path / "ds"
    (data / "dep_index").mkdir(parents=True)
    (tmp_path / "data").mkdir()
    poses = [np.eye(4)]
    for k in range(4):
        poses.append(step @ poses[-1])
    hb = max(2, int(round(32 * W / 640.0)))
    for k, P in enumerate(poses):
        d = synth.render_depth(P, intr, 4242, 10 + k, hole_block=hb)
        _write_png16(str(data / "dep_index" / f"{k + 1}.png"), d)
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=10.0, fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=W, h=H, lc="no", planes="no", pcd="no", extra=""))
    out = subprocess.run([os.path.join(HOST, "run_SLAM"), "4"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    norms = [float(x) for x in (tmp_path / "data" / "error_of_transform.log").read_text().split()]
    assert len(norms) == 4 and all(0 < n < 0.5 for n in norms) and norms == sorted(norms)
    traj = np.loadtxt(str(tmp_path / "data" / "trajectory_icp.txt"))
    assert traj.shape == (5, 8)
    for k in range(1, 5):
        cam_to_world = np.linalg.inv(poses[k])          # robot = T^-1 * kf_pos with kf_pos = I (src/GraphicEnd.cpp:169-170,245)
        assert np.abs(traj[k, 1:4] - cam_to_world[:3, 3]).max() < 3e-2   # quarter-resolution ICP accuracy, not a parity bar
    assert (tmp_path / "data" / "keyframe.txt").read_text().split() == ["0", "1"]


def _quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _se3(v):
    T = np.eye(4)
    T[:3, 3] = v[:3]
    T[:3, :3] = _quat_to_R(v[3:7])
    return T


@pytest.mark.gpu
def test_run_slam_keyframes_loop_closure_and_g2o_handoff(gpu_lib, tmp_path):
    """Rows f-3 / f-4: a there-and-back synthetic sequence with a small max_pos_change makes every frame a keyframe
    (src/GraphicEnd.cpp:230-240); loopClosure (:685-762) aligns the new keyframe against the adjacent and the
    random earlier keyframes in ONE batched launch and adds EdgeSE3s; saveFinalResult writes the graph in g2o
    text format (:661-682) and keyframe.txt; planes.txt holds the per-frame plane list (:353-430)."""
    _build_host()
    W, H = 320, 240
    intr = synth.Intrinsics.scaled(W, H)
    step = synth.pose_from_seed(4242, max_angle_deg=1.0, max_trans=0.02)
    data = tmp_path / "ds"
    (data / "dep_index").mkdir(parents=True)
    (tmp_path / "data").mkdir()
    out_poses = [np.eye(4)]
    for k in range(4):
        out_poses.append(step @ out_poses[-1])
    poses = out_poses + out_poses[-2::-1]                      # 0 1 2 3 4 3 2 1 0
    hb = max(2, int(round(32 * W / 640.0)))
    (data / "pcd").mkdir()
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
            "WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    voxels, pcd_clouds = [], []
    for k, P in enumerate(poses):
        d = synth.render_depth(P, intr, 4242, 10 + k, hole_block=hb)
        _write_png16(str(data / "dep_index" / f"{k + 1}.png"), d)
        c = synth.backproject_numpy(d, intr, z_filter=1e9).reshape(-1, 4)
        c = c[np.isfinite(c[:, 2])].copy()                      # convert2PCD drops d == 0 (src/convert2PCD.cpp:60-61)
        c[:, 3] = np.float32(0)
        (data / "pcd" / f"{k + 1}.pcd").write_bytes(head.format(n=c.shape[0]).encode() + c.tobytes())
        voxels.append((c.shape[0], O.voxel_grid(c, 0.03, 7.0).shape[0]))
        pcd_clouds.append(c)
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=0.005, fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=W, h=H, lc="yes", planes="yes", pcd="yes", extra="icp_plane_gate: yes\nhip_devices: 2\nhip_devices_share: yes"))
    out = subprocess.run([os.path.join(HOST, "run_SLAM"), str(len(poses) - 1)], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    kf = np.loadtxt(str(tmp_path / "data" / "keyframe.txt"), dtype=int)
    assert kf.shape == (len(poses), 2) and list(kf[:, 0]) == list(range(len(poses)))       # every frame a keyframe
    V, E, fixed = {}, [], []
    for line in (tmp_path / "data" / "final.g2o").read_text().splitlines():
        t = line.split()
        if t[0] == "VERTEX_SE3:QUAT":
            V[int(t[1])] = _se3([float(x) for x in t[2:9]])
        elif t[0] == "EDGE_SE3:QUAT":
            assert len(t) == 3 + 7 + 21
            info = [float(x) for x in t[10:]]
            assert info == [100, 0, 0, 0, 0, 0, 100, 0, 0, 0, 0, 100, 0, 0, 0, 100, 0, 0, 100, 0, 100]   # :330-334
            E.append((int(t[1]), int(t[2]), _se3([float(x) for x in t[3:10]])))
        elif t[0] == "FIX":
            fixed.append(int(t[1]))
    assert sorted(V) == list(range(len(poses))) and fixed == [0]
    odo = [(a, b) for a, b, _ in E if b == a + 1]
    loops = [(a, b) for a, b, _ in E if b != a + 1]
    assert len(set(odo)) == len(poses) - 1 and len(loops) >= 4                               # chain + closures
    for a, b, M in E:                                        # every edge agrees with the vertex estimates
        err = np.linalg.inv(M) @ np.linalg.inv(V[a]) @ V[b]
        ang = np.arccos(np.clip((np.trace(err[:3, :3]) - 1) / 2, -1, 1))
        assert ang < 0.03 and np.linalg.norm(err[:3, 3]) < 0.06, (a, b, ang, err[:3, 3])
    # the camera came back: last vertex close to the first
    assert np.linalg.norm(V[len(poses) - 1][:3, 3]) < 0.06
    # row f-1: every frame's PCD went through PassThrough + VoxelGrid(grid_leaf) on the GPU
    clouds = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("cloud ")]
    assert [(int(t[1]), int(t[3])) for t in clouds] == voxels
    lc = (tmp_path / "data" / "lc.txt").read_text().split()
    assert len(lc) % 4 == 0
    planes = (tmp_path / "data" / "planes.txt").read_text().strip().splitlines()
    assert len(planes) == len(poses)
    for ln in planes:
        t = ln.split()
        n = int(t[1])
        assert 2 <= n <= 3 and len(t) == 2 + 5 * n and all(float(t[2 + 5 * k + 3]) >= 0 for k in range(n))   # d >= 0 (:383-387)

    # row f-3: the reference's map builder (src/saveOutput.cpp) on the files just written: VoxelGrid, PassThrough z <= 5,
    # transform by the g2o vertex pose, merge, VoxelGrid -> result.pcd; checked against the oracle pipeline
    out = subprocess.run([os.path.join(HOST, "saveOutput"), "data/keyframe.txt", "data/final.g2o"], cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    parts = []
    for kid, frame in kf:
        v = O.voxel_grid_only(pcd_clouds[frame - 1], 0.03)
        t, _ = O.pass_transform(v, V[kid], 5.0)
        parts.append(t)
    want = O.voxel_grid_only(np.concatenate(parts), 0.03)
    txt = (tmp_path / "result.pcd").read_text().splitlines()
    assert txt[2] == "FIELDS x y z rgba" and txt[10] == "DATA ascii"
    got = np.array([[float(x) for x in ln.split()[:3]] for ln in txt[11:]], dtype=np.float32)
    assert f"POINTS {got.shape[0]}" == txt[9]
    # the poses went through text (9 significant digits) in both pipelines but are rebuilt by different code: a
    # voxel on a face can flip, nothing more
    assert abs(got.shape[0] - want.shape[0]) <= max(3, want.shape[0] // 2000)
    if got.shape[0] == want.shape[0]:
        assert np.abs(got - want[:, :3]).max() < 0.035
    assert "final result saved" in out.stdout


@pytest.mark.gpu
def test_two_threads_two_handles_on_one_device_equal_one_handle(gpu_lib, tmp_path):
    """GraphicEndICP::multiPnPBatch shards a loop-closure batch over its handles, one host THREAD per handle.
    With `hip_devices: 2` + `hip_devices_share: yes` both handles live on this GPU and share the device's run counter (the only
    state handles of one device share; library-owned, atomics only, speed only).  The run must produce byte-identical lc.txt,
    final.g2o and error log to the one-handle run: sharding and threading change no result."""
    _build_host()
    step = synth.pose_from_seed(4243, max_angle_deg=1.0, max_trans=0.02)
    out_poses = [np.eye(4)]
    for k in range(4):
        out_poses.append(step @ out_poses[-1])
    poses = out_poses + out_poses[-2::-1]
    outs = {}
    for name, extra in (("one", ""), ("two", "hip_devices: 2\nhip_devices_share: yes\n")):
        d = tmp_path / name
        d.mkdir()
        intr, data = _sequence(d, poses)
        (d / "parameters.yaml").write_text(
            PARAMS.format(src=str(data), mpc=0.005, fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="yes", planes="no", pcd="no", extra=extra))
        out = subprocess.run([os.path.join(HOST, "run_SLAM"), str(len(poses) - 1)], cwd=str(d), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert f"ICP front end on {1 if name == 'one' else 2} GPU(s)" in out.stdout
        outs[name] = {f: (d / "data" / f).read_text() for f in ("lc.txt", "final.g2o", "error_of_transform.log", "keyframe.txt")}
    assert len(outs["one"]["lc.txt"].split()) >= 8                  # closures were found (several pairs per batch: both threads had work)
    for f in outs["one"]:
        assert outs["one"][f] == outs["two"][f], f


def _sequence(tmp_path, poses, W=320, H=240, seed=4242, blank=()):
    intr = synth.Intrinsics.scaled(W, H)
    data = tmp_path / "ds"
    (data / "dep_index").mkdir(parents=True)
    (tmp_path / "data").mkdir()
    hb = max(2, int(round(32 * W / 640.0)))
    for k, P in enumerate(poses):
        d = synth.render_depth(P, intr, seed, 10 + k, hole_block=hb)
        if k in blank:
            d = np.zeros_like(d)                                    # a frame with no valid depth: cannot be aligned
        _write_png16(str(data / "dep_index" / f"{k + 1}.png"), d)
    return intr, data


@pytest.mark.gpu
def test_run_slam_lost_frames_and_lost_recovery(gpu_lib, tmp_path):
    """The lost branch of GraphicEnd::run (src/GraphicEnd.cpp:173-229) and lostRecovery (:764-838): frames 4, 5, 6 carry
    no depth -> each aligns to neither the keyframe nor the last frame -> '9999' in error_of_transform.log and _lost++;
    with lost_frames = 2 the third lost frame triggers 'Lost Recovery': the present frame becomes a keyframe without an
    odometry edge (lost.txt, keyframe.txt).  That keyframe is blank too, so frames 7..9 are lost again and frame 9 (valid)
    becomes the next recovery keyframe, against which frames 10, 11 track."""
    _build_host()
    step = synth.pose_from_seed(4242, max_angle_deg=1.0, max_trans=0.02)
    poses = [np.eye(4)]
    for k in range(10):
        poses.append(step @ poses[-1])
    intr, data = _sequence(tmp_path, poses, blank=(3, 4, 5))
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=10.0, fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="no", planes="no", pcd="no",
                      extra="").replace("lost_frames: 10", "lost_frames: 2"))
    out = subprocess.run([os.path.join(HOST, "run_SLAM"), "10"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    log = (tmp_path / "data" / "error_of_transform.log").read_text().split()
    assert len(log) == 10
    assert log[2:8] == ["9999"] * 6                                 # frames 4..9: no depth, then a blank keyframe (:176)
    assert all(0 < float(x) < 0.5 for x in log[:2] + log[8:])       # frames 2, 3 track keyframe 0; frames 10, 11 track keyframe 9
    assert out.stdout.count("This frame lost") == 6 and out.stdout.count("Lost Recovery...") == 2
    lost = (tmp_path / "data" / "lost.txt").read_text().split()
    assert lost == ["1", "6", "2", "9"]                             # "keyframe id, frame index" per recovery (:774-776)
    kf = np.loadtxt(str(tmp_path / "data" / "keyframe.txt"), dtype=int).reshape(-1, 2)
    assert kf.tolist() == [[0, 1], [1, 6], [2, 9]]
    edges = [ln.split() for ln in (tmp_path / "data" / "final.g2o").read_text().splitlines() if ln.startswith("EDGE_SE3")]
    # recovery keyframes get no odometry edge (:793); lostRecovery aligns the new keyframe against every earlier one
    # (:808-836): the blank keyframe 1 matches nothing, keyframe 2 (frame 9) finds keyframe 0 (frame 1) again
    assert [(e[1], e[2]) for e in edges] == [("0", "2")]


@pytest.mark.gpu
def test_run_slam_last_frame_becomes_keyframe(gpu_lib, tmp_path):
    """The other half of the lost branch (src/GraphicEnd.cpp:186-228): the present frame fails against the keyframe
    (here: norm above error_threshold, :621) but matches the LAST ordinary frame -> last becomes a keyframe (stamped
    with its own frame index, _index - 1, :198) and then the present one (:227)."""
    _build_host()
    step = synth.pose_from_seed(4242, max_angle_deg=1.0, max_trans=0.02)
    ang = np.arccos(np.clip((np.trace(step[:3, :3]) - 1) / 2, -1, 1))
    norm_step = ang + 0.9 * np.linalg.norm(step[:3, 3])
    poses = [np.eye(4)]
    for k in range(6):
        poses.append(step @ poses[-1])
    intr, data = _sequence(tmp_path, poses)
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=10.0, fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="no", planes="no", pcd="no",
                      extra="").replace("error_threshold: 1.0", f"error_threshold: {1.5 * norm_step:.6f}"))
    out = subprocess.run([os.path.join(HOST, "run_SLAM"), "6"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    log = (tmp_path / "data" / "error_of_transform.log").read_text().split()
    # frame 2: one step from keyframe 0 (ok); frame 3: two steps (> 1.5 steps: lost, but one step from frame 2) -> keyframes
    # frame 2 and frame 3; frame 4: one step from keyframe frame 3 (ok); frame 5: lost again -> keyframes 4 and 5; ...
    assert log[1] == "9999" and log[3] == "9999" and 0 < float(log[0]) < 1.5 * norm_step and 0 < float(log[2]) < 1.5 * norm_step
    kf = np.loadtxt(str(tmp_path / "data" / "keyframe.txt"), dtype=int).reshape(-1, 2)
    assert kf[:5].tolist() == [[0, 1], [1, 2], [2, 3], [3, 4], [4, 5]]            # the first of each pair carries _index - 1
    assert "lost.txt" not in os.listdir(str(tmp_path / "data"))                    # never lost for more than one frame
    # the chained vertex poses follow the ground truth
    V = {}
    for line in (tmp_path / "data" / "final.g2o").read_text().splitlines():
        t = line.split()
        if t[0] == "VERTEX_SE3:QUAT":
            V[int(t[1])] = _se3([float(x) for x in t[2:9]])
    for kid, frame in kf:
        err = poses[frame - 1] @ V[kid]                             # world->camera (truth) times camera->world (estimate)
        assert np.linalg.norm(err[:3, 3]) < 0.05 and np.arccos(np.clip((np.trace(err[:3, :3]) - 1) / 2, -1, 1)) < 0.02


@pytest.mark.gpu
def test_trajectory_over_several_rotated_keyframes(gpu_lib, tmp_path):
    """camera-to-world poses chain as kf_pos * T over several keyframes with real rotation; the TUM lines of
    trajectory_icp.txt and the g2o vertices must follow the ground truth, not only for kf_pos = I."""
    _build_host()
    step = synth.pose_from_seed(99, max_angle_deg=3.0, max_trans=0.05)
    poses = [np.eye(4)]
    for k in range(6):
        poses.append(step @ poses[-1])
    intr, data = _sequence(tmp_path, poses, seed=99)
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=0.03, fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="no", planes="no", pcd="no", extra=""))
    out = subprocess.run([os.path.join(HOST, "run_SLAM"), "6"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    kf = np.loadtxt(str(tmp_path / "data" / "keyframe.txt"), dtype=int).reshape(-1, 2)
    assert len(kf) >= 3                                               # several keyframes, each rotated w.r.t. the first
    traj = np.loadtxt(str(tmp_path / "data" / "trajectory_icp.txt"))
    assert traj.shape == (7, 8)
    for k in range(1, 7):
        want = np.linalg.inv(poses[k])                                # camera-to-world of frame k
        got = _se3(traj[k, 1:8])
        err = np.linalg.inv(want) @ got
        ang = np.arccos(np.clip((np.trace(err[:3, :3]) - 1) / 2, -1, 1))
        assert ang < 0.02 and np.linalg.norm(err[:3, 3]) < 0.05, (k, ang, err[:3, 3])


def test_plane_gate_twin_without_gpu():
    """slam3d_plane_gate (host code of the library, what GraphicEndICP::planeGate does through match()): planes of
    frame 1 carried into frame 2 by T and matched on (a, b, c, d); numpy twin."""
    from slam3d_gx_amd import capi
    rng = np.random.default_rng(5)

    def unit_planes(n):
        p = rng.normal(size=(n, 4))
        p[:, :3] /= np.linalg.norm(p[:, :3], axis=1, keepdims=True)
        p[:, 3] = np.abs(p[:, 3]) + 0.5
        return p.astype(np.float32)

    def moved(p, T):
        n = p[:, :3].astype(np.float64) @ T[:3, :3].T
        d = p[:, 3].astype(np.float64) - n @ T[:3, 3]
        s = np.where(d < 0, -1.0, 1.0)
        return np.concatenate([n * s[:, None], (d * s)[:, None]], axis=1).astype(np.float32)

    for trial in range(20):
        T = synth.pose_from_seed(100 + trial, max_angle_deg=20.0, max_trans=0.5)
        p1 = unit_planes(3)
        p2 = moved(p1, T)[[2, 0]] + rng.normal(scale=0.01, size=(2, 4)).astype(np.float32)      # two of them seen again
        assert capi.plane_gate(p1, p2, T, 0.15) == 2
        assert capi.plane_gate(p1, p2, np.eye(4), 0.02) == int(sum(
            np.linalg.norm(moved(p1, np.eye(4))[:, None, :].astype(np.float64) - p2[None], axis=2).min(1) <= 0.02))
        wrong = synth.pose_from_seed(500 + trial, max_angle_deg=60.0, max_trans=1.5)
        d = np.linalg.norm(moved(p1, wrong)[:, None, :].astype(np.float64) - p2[None].astype(np.float64), axis=2).min(1)
        assert capi.plane_gate(p1, p2, wrong, 0.15) == int((d <= 0.15).sum())
    assert capi.plane_gate(np.zeros((0, 4), np.float32), unit_planes(2), np.eye(4)) == 0


@pytest.mark.gpu
def test_control_flow_against_python_twin(gpu_lib, tmp_path):
    """Row f-4 with an ORACLE for the control flow (the loop-closure test was property-only): tests/
    slam_twin.py re-states run / generateKeyFrame / loopClosure / lostRecovery / findMoreLoops in Python over the same
    C-ABI.  A there-and-back sequence with two blank frames (lost branch, recovery) and loop-closure detection on must
    give the SAME keyframes, lost list, log lines, loop-closure lines and graph edges from run_SLAM and from the twin."""
    import slam_twin
    _build_host()
    step = synth.pose_from_seed(4242, max_angle_deg=1.0, max_trans=0.02)
    fwd = [np.eye(4)]
    for k in range(5):
        fwd.append(step @ fwd[-1])
    poses = fwd + fwd[-2::-1] + fwd[1:4]                             # 0..5, back to 0, out again to 3: 14 frames
    blank = (7, 8)
    intr, data = _sequence(tmp_path, poses, blank=blank)
    cfg = dict(max_pos_change=0.005, lost_frames=0, loop_closure_detection=True, loopclosure_frames=4, icp_iterations=15)
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=cfg["max_pos_change"], fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="yes", planes="no",
                      pcd="no", extra="").replace("lost_frames: 10", "lost_frames: 0"))
    n = len(poses) - 1
    out = subprocess.run([os.path.join(HOST, "run_SLAM"), str(n)], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]

    from PIL import Image
    depth_of = lambda i: np.array(Image.open(str(data / "dep_index" / f"{i}.png"))).astype(np.uint16)
    tw = slam_twin.Twin(intr, depth_of, cfg)
    try:
        for _ in range(n):
            tw.run()
        tw.find_more_loops()
    finally:
        tw.close()

    kf = np.loadtxt(str(tmp_path / "data" / "keyframe.txt"), dtype=int).reshape(-1, 2)
    assert kf.tolist() == [[k["id"], k["frame_index"]] for k in tw.keyframes]
    log = (tmp_path / "data" / "error_of_transform.log").read_text().split()
    assert len(log) == len(tw.err_log)
    for a, b in zip(log, tw.err_log):
        assert a == b if isinstance(b, str) else abs(float(a) - b) <= 1e-5 * max(1.0, abs(b)), (a, b)
    assert "9999" in log                                              # the blank frames took the lost branch
    lost_path = tmp_path / "data" / "lost.txt"
    lost = [int(x) for x in lost_path.read_text().split()] if lost_path.exists() else []
    assert lost == [v for pair in tw.lost_lines for v in pair] and len(lost) >= 2
    lc = [ln.split() for ln in (tmp_path / "data" / "lc.txt").read_text().splitlines()]
    assert [(int(t[0]), int(t[1]), int(t[3])) for t in lc] == [(a, b, d) for a, b, _, d in tw.lc_lines]
    for t, (_, _, nrm, _) in zip(lc, tw.lc_lines):
        assert abs(float(t[2]) - nrm) <= 1e-5 * max(1.0, nrm)
    assert len(lc) >= 3
    edges = [(int(t[1]), int(t[2])) for t in (ln.split() for ln in (tmp_path / "data" / "final.g2o").read_text().splitlines()) if t[0] == "EDGE_SE3:QUAT"]
    assert edges == tw.edges
    traj = np.loadtxt(str(tmp_path / "data" / "trajectory_icp.txt"))
    assert traj.shape[0] == len(tw.traj)
    for row, (idx, T) in zip(traj, tw.traj):
        assert int(row[0]) == idx and np.abs(row[1:4] - T[:3, 3]).max() < 1e-6


@pytest.mark.gpu
def test_correspondence_gates_reach_the_device_from_parameters_yaml(gpu_lib, tmp_path):
    """Rows a8 / a11: `icp_plane_residual_gate: yes` applies the reference's own `min_error_plane` key to the squared
    point-to-plane residual, `icp_normal_angle_deg` the normal-angle gate; run_SLAM with both == the Python twin over the
    C-ABI with the same two numbers, and the gated log differs from the ungated one (the gates bite)."""
    import slam_twin
    _build_host()
    step = synth.pose_from_seed(777, max_angle_deg=1.0, max_trans=0.02)
    poses = [np.eye(4)]
    for k in range(4):
        poses.append(step @ poses[-1])
    intr, data = _sequence(tmp_path, poses)
    r2, deg = 1e-5, 20.0
    cfg = dict(max_pos_change=0.005, icp_iterations=15)
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=cfg["max_pos_change"], fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="no", planes="no",
                      pcd="no", extra=f"icp_plane_residual_gate: yes\nmin_error_plane: {r2}\nicp_normal_angle_deg: {deg}\n"))
    n = len(poses) - 1
    out = subprocess.run([os.path.join(HOST, "run_SLAM"), str(n)], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    log = [float(x) for x in (tmp_path / "data" / "error_of_transform.log").read_text().split()]

    from PIL import Image
    depth_of = lambda i: np.array(Image.open(str(data / "dep_index" / f"{i}.png"))).astype(np.uint16)
    logs = {}
    for name, extra in (("gated", dict(max_plane_residual2=r2, min_normal_cos=float(np.cos(np.deg2rad(deg))))), ("plain", {})):
        tw = slam_twin.Twin(intr, depth_of, dict(cfg, **extra))
        try:
            for _ in range(n):
                tw.run()
        finally:
            tw.close()
        logs[name] = [float(x) for x in tw.err_log]
    assert len(log) == len(logs["gated"]) == n
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(log, logs["gated"])), (log, logs["gated"])
    assert any(abs(a - b) > 1e-7 for a, b in zip(logs["gated"], logs["plain"]))


@pytest.mark.gpu
def test_run_slam_aligns_the_voxel_clouds_readimage_makes(gpu_lib, tmp_path):
    """the reference hands the cloud readimage produced -- PCD -> PassThrough -> VoxelGrid(0.03) -- on to its
    alignment (src/GraphicEnd.cpp:279-295 -> :158).  `icp_cloud: voxel` makes GraphicEndICP::multiPnP align exactly those point lists
    (a height-1 handle, svd, the persistent list launch): run_SLAM on PCD input == the Python twin that aligns the ORACLE's voxel
    clouds through the C-ABI, and differs from the depth-image run of the same sequence."""
    import slam_twin
    _build_host()
    step = synth.pose_from_seed(4242, max_angle_deg=1.0, max_trans=0.02)
    poses = [np.eye(4)]
    for k in range(4):
        poses.append(step @ poses[-1])
    intr, data = _sequence(tmp_path, poses)
    (data / "pcd").mkdir()
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
            "WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    from PIL import Image
    depth_of = lambda i: np.array(Image.open(str(data / "dep_index" / f"{i}.png"))).astype(np.uint16)
    vox = {}
    for k in range(len(poses)):
        c = synth.backproject_numpy(depth_of(k + 1), intr, z_filter=1e9).reshape(-1, 4)
        c = c[np.isfinite(c[:, 2])].copy()                      # convert2PCD drops d == 0 (src/convert2PCD.cpp:60-61)
        c[:, 3] = np.float32(0)
        (data / "pcd" / f"{k + 1}.pcd").write_bytes(head.format(n=c.shape[0]).encode() + c.tobytes())
        vox[k + 1] = O.voxel_grid(c, 0.03, 7.0)
    assert all(10000 < len(v) < 32768 for v in vox.values()), [len(v) for v in vox.values()]      # > 384 tiles: the boxes do not fit LDS, read from L2
    cfg = dict(max_pos_change=0.005, icp_iterations=15)
    n = len(poses) - 1
    logs = {}
    for mode in ("voxel", "depth"):
        extra = "icp_cloud: voxel\n" if mode == "voxel" else ""
        (tmp_path / "parameters.yaml").write_text(
            PARAMS.format(src=str(data), mpc=cfg["max_pos_change"], fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="no", planes="no",
                          pcd="yes", extra=extra))
        out = subprocess.run([os.path.join(HOST, "run_SLAM"), str(n)], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        logs[mode] = [float(x) for x in (tmp_path / "data" / "error_of_transform.log").read_text().split()]
        if mode == "voxel":
            got = [int(t.split()[2]) for t in out.stdout.splitlines() if t.startswith("multiICP::inliers")]      # "inliers = a / n_src"
            nsrc = [int(t.split()[4].rstrip(",")) for t in out.stdout.splitlines() if t.startswith("multiICP::inliers")]
            assert got and all(a > 1000 for a in got) and set(nsrc) <= {len(v) for v in vox.values()}, (got, nsrc)   # the sources ARE the voxel clouds
    tw = slam_twin.Twin(intr, depth_of, dict(cfg, icp_cloud="voxel"), cloud_of=lambda i: vox[i])
    try:
        for _ in range(n):
            tw.run()
    finally:
        tw.close()
    want = [float(x) for x in tw.err_log]
    assert len(logs["voxel"]) == len(want) == n
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(logs["voxel"], want)), (logs["voxel"], want)
    assert any(abs(a - b) > 1e-7 for a, b in zip(logs["voxel"], logs["depth"]))
    # icp_cloud: voxel without PCD input is a fatal configuration error, like the reference's (src/GraphicEnd.cpp:113)
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=0.005, fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="no", planes="no", pcd="no", extra="icp_cloud: voxel\n"))
    bad = subprocess.run([os.path.join(HOST, "run_SLAM"), "1"], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "icp_read_pcd" in bad.stderr


@pytest.mark.gpu
def test_plane_estimator_reaches_the_device_from_parameters_yaml(gpu_lib, tmp_path):
    """`icp_estimator: plane` (+ `icp_plane_pair_gate: yes`): run_SLAM aligns with SLAM3D_EST_PLANE -- the planes the library extracts
    per frame give the normals (src/GraphicEnd.cpp:158,168: planes per frame drive the pose), correspondences only inside associated
    plane pairs (:459-484,:572) -- and logs exactly what the Python twin gets over the C-ABI with the same parameters; the log
    differs from the window-normal estimator's."""
    import slam_twin
    from slam3d_gx_amd import capi
    _build_host()
    step = synth.pose_from_seed(778, max_angle_deg=1.0, max_trans=0.02)
    poses = [np.eye(4)]
    for k in range(4):
        poses.append(step @ poses[-1])
    intr, data = _sequence(tmp_path, poses)
    cfg = dict(max_pos_change=0.005, icp_iterations=15)
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=cfg["max_pos_change"], fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="no", planes="no",
                      pcd="no", extra="icp_estimator: plane\nicp_plane_pair_gate: yes\n"))
    n = len(poses) - 1
    out = subprocess.run([os.path.join(HOST, "run_SLAM"), str(n)], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    log = [float(x) for x in (tmp_path / "data" / "error_of_transform.log").read_text().split()]
    from PIL import Image
    depth_of = lambda i: np.array(Image.open(str(data / "dep_index" / f"{i}.png"))).astype(np.uint16)
    logs = {}
    for name, extra in (("plane", dict(estimator=capi.EST_PLANE, plane_flags=capi.PLANE_PAIR_GATE)), ("window", {})):
        tw = slam_twin.Twin(intr, depth_of, dict(cfg, **extra))
        try:
            for _ in range(n):
                tw.run()
        finally:
            tw.close()
        logs[name] = [float(x) for x in tw.err_log]
    assert len(log) == len(logs["plane"]) == n
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(log, logs["plane"])), (log, logs["plane"])
    assert any(abs(a - b) > 1e-7 for a, b in zip(logs["plane"], logs["window"]))


@pytest.mark.gpu
def test_motion_model_initial_guess_matches_the_twin_and_tracks(gpu_lib, tmp_path):
    """`icp_motion_model: yes`: the previous frame's pose against the same keyframe starts the present frame's alignment
    (reset at every keyframe change).  run_SLAM == the Python twin with the same rule, and the trajectory still follows
    the ground truth."""
    import slam_twin
    _build_host()
    step = synth.pose_from_seed(91, max_angle_deg=0.8, max_trans=0.015)
    poses = [np.eye(4)]
    for k in range(7):
        poses.append(step @ poses[-1])
    intr, data = _sequence(tmp_path, poses)
    cfg = dict(max_pos_change=0.06, icp_iterations=15, icp_motion_model=True)       # a keyframe every few frames
    (tmp_path / "parameters.yaml").write_text(
        PARAMS.format(src=str(data), mpc=cfg["max_pos_change"], fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=320, h=240, lc="no", planes="no",
                      pcd="no", extra="icp_motion_model: yes\n"))
    n = len(poses) - 1
    out = subprocess.run([os.path.join(HOST, "run_SLAM"), str(n)], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    log = [float(x) for x in (tmp_path / "data" / "error_of_transform.log").read_text().split()]
    from PIL import Image
    depth_of = lambda i: np.array(Image.open(str(data / "dep_index" / f"{i}.png"))).astype(np.uint16)
    tw = slam_twin.Twin(intr, depth_of, cfg)
    try:
        for _ in range(n):
            tw.run()
    finally:
        tw.close()
    assert len(log) == len(tw.err_log) == n
    assert all(abs(a - float(b)) <= 1e-5 * max(1.0, abs(float(b))) for a, b in zip(log, tw.err_log)), (log, tw.err_log)
    kf = np.loadtxt(str(tmp_path / "data" / "keyframe.txt"), dtype=int).reshape(-1, 2)
    assert kf.tolist() == [[k["id"], k["frame_index"]] for k in tw.keyframes] and 2 <= len(kf) < n
    traj = np.loadtxt(str(tmp_path / "data" / "trajectory_icp.txt"))
    # the camera's pose in frame-0 coordinates is the inverse of the scene motion the sequence was rendered with
    for row, P in zip(traj[1:], poses[1:]):
        assert np.abs(row[1:4] - np.linalg.inv(P)[:3, 3]).max() < 5e-3


@pytest.mark.gpu
def test_planar_features_binary_on_config1_frame(gpu_lib, tmp_path):
    """BASELINE config 1's binary (src/planarFeatures.cpp:26-136, minus OpenCV's window and FAST): `planarFeatures dep.png`
    on the reference's bin/dep_1.png (tests/golden/kinect) through k_normals must count exactly what the oracle's spec S2
    and a numpy reading of the reference's "no zero in the patch" rule give -- for every pixel, and for a key point list."""
    _build_host()
    from PIL import Image
    png = os.path.join(ROOT, "tests", "golden", "kinect", "bin_dep_1.png")
    dep = np.array(Image.open(png)).astype(np.uint16)
    H, W = dep.shape
    intr = synth.Intrinsics(W, H, 525.0, 525.0, 320.0, 235.5, 1000.0)           # src/planarFeatures.cpp:13-14
    p = O.params(intr, z_filter=10.0)
    nrm = O.normals(O.backproject(dep, p), p)
    planar = nrm[..., 3] > 0.5
    inner = np.zeros_like(planar); inner[3:H - 3, 3:W - 3] = True
    from numpy.lib.stride_tricks import sliding_window_view
    nozero = np.zeros_like(planar)
    nozero[3:H - 3, 3:W - 3] = (sliding_window_view(dep, (7, 7)) != 0).all(axis=(2, 3))
    exe = os.path.join(HOST, "planarFeatures")
    out = subprocess.run([exe, png], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    want = f"total kp: {int(inner.sum())}, valid: {int((inner & (dep != 0)).sum())}, planar: {int((inner & planar).sum())}"
    assert want in out.stdout, (want, out.stdout)
    assert f"(the reference's rule): {int((inner & planar & nozero).sum())}" in out.stdout, out.stdout
    assert int((inner & planar).sum()) > 50000                                   # the frame is mostly floor and walls
    # a key point list (what cv::FAST would hand over), truncated like the reference does
    rng = np.random.default_rng(5)
    kps = np.stack([rng.uniform(0, W, 500), rng.uniform(0, H, 500)], 1)
    kp_file = tmp_path / "kp.txt"
    kp_file.write_text("\n".join(f"{u:.3f} {v:.3f}" for u, v in kps))
    out = subprocess.run([exe, png, str(kp_file)], capture_output=True, text=True, timeout=300)
    ui, vi = kps[:, 0].astype(int), kps[:, 1].astype(int)
    ok = (ui >= 3) & (vi >= 3) & (ui + 3 < W) & (vi + 3 < H)
    v_ = ok & (dep[np.clip(vi, 0, H - 1), np.clip(ui, 0, W - 1)] != 0)
    pl = v_ & planar[np.clip(vi, 0, H - 1), np.clip(ui, 0, W - 1)]
    assert f"total kp: 500, valid: {int(v_.sum())}, planar: {int(pl.sum())}" in out.stdout, out.stdout

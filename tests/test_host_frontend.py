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
loop_closure_detection: no
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
    yml.write_text(PARAMS.format(src="/data/x", mpc=0.25, fx=517.0, fy=517.0, cx=318.6, cy=255.3, w=320, h=240))
    out = subprocess.run([os.path.join(HOST, "host_selftest"), str(png), str(yml)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    flat = pr.depth_src.reshape(-1).astype(np.uint64)
    want = int((flat * (np.arange(flat.size, dtype=np.uint64) % np.uint64(9973) + np.uint64(1))).sum())
    assert f"png 320 240 {want} {int((flat != 0).sum())}" in lines
    # GetPara returns strings; unknown key -> "unknown_para_name" + stderr (src/ParameterReader.cpp:121-122)
    assert "data_source=/data/x z_filter=7.0 max_planes=3 missing=unknown_para_name fx=517.000 factor=1000.0 iters=15" in lines[-1]
    assert "Unknown parameter: no_such_key" in out.stderr


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
        PARAMS.format(src=str(data), mpc=10.0, fx=intr.fx, fy=intr.fy, cx=intr.cx, cy=intr.cy, w=W, h=H))
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

"""The C-ABI library loads on a CPU-only host and exports every symbol include/slam3d_icp.h declares."""
import ctypes
import os
import re

import pytest

from slam3d_gx_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "slam3d_icp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(slam3d_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    so = build.build_lib()
    lib = ctypes.CDLL(so)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/slam3d_icp.h but not exported"
    assert sorted(capi.EXPORTED_SYMBOLS) == declared


def test_abi_version_and_strerror_without_gpu():
    lib = capi.load_library()
    assert lib.slam3d_icp_abi_version() == 8
    assert b"no gfx950" in lib.slam3d_strerror(-4)
    assert lib.slam3d_strerror(0) == b"ok"


def test_struct_layouts_match_header_sizes(tmp_path):
    """Compile the public header with plain gcc (it must be valid C) and compare struct sizes/offsets
    with the ctypes mirror used by tests and bench."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "slam3d_icp.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(slam3d_icp_params), sizeof(slam3d_icp_result),'
        ' sizeof(slam3d_cloud_view), sizeof(slam3d_plane), offsetof(slam3d_icp_params, nn_mode),'
        ' offsetof(slam3d_icp_result, rmse), offsetof(slam3d_icp_result, T_raw), sizeof(slam3d_seg_params),'
        ' offsetof(slam3d_seg_params, seed), offsetof(slam3d_icp_params, extra_frames), sizeof(slam3d_pose_record),'
        ' offsetof(slam3d_pose_record, rmse));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(capi.Params), ctypes.sizeof(capi.Result), ctypes.sizeof(capi.CloudView),
                   ctypes.sizeof(capi.Plane), capi.Params.nn_mode.offset, capi.Result.rmse.offset, capi.Result.T_raw.offset,
                   ctypes.sizeof(capi.SegParams), capi.SegParams.seed.offset, capi.Params.extra_frames.offset,
                   ctypes.sizeof(capi.PoseRecord), capi.PoseRecord.rmse.offset]
    assert ctypes.sizeof(capi.PoseRecord) == 160       # 160-byte pose records


def test_shard_range_and_comm_entry_points_without_gpu():
    """The multi-GPU entry points are exported and the ones that need no device behave: contiguous pair / row
    blocks, remainder to the lowest ranks; the communicator refuses bad arguments."""
    for n, world in ((64, 8), (10, 3), (3, 8), (0, 4), (480, 7)):
        cover = []
        for r in range(world):
            b, e = capi.shard_range(n, world, r)
            assert 0 <= b <= e <= n
            cover += list(range(b, e))
        assert cover == list(range(n))
    assert capi.shard_range(10, 3, 0) == (0, 4) and capi.shard_range(10, 3, 2) == (7, 10)
    lib = capi.load_library()
    out = ctypes.c_void_p()
    assert lib.slam3d_comm_init(None, 0, 1, 0, ctypes.byref(out)) == -1          # SLAM3D_E_INVALID
    buf = (ctypes.c_ubyte * 128)()
    assert lib.slam3d_comm_init(buf, 3, 2, 0, ctypes.byref(out)) == -1           # rank >= world


def test_create_fails_loudly_without_device_or_with_bad_params():
    import torch
    p = capi.default_params()
    if not torch.cuda.is_available():
        with pytest.raises(capi.Slam3dError) as e:
            capi.IcpHandle(p)
        assert e.value.code == -4          # SLAM3D_E_NODEVICE: no CPU fallback
    p.normal_window = 4
    with pytest.raises(capi.Slam3dError) as e:
        capi.IcpHandle(p)
    assert e.value.code == -1
    p = capi.default_params()
    p.z_filter = 500.0                      # 640x480 points at up to ~600 m: the int64 fixed-point sums could overflow
    with pytest.raises(capi.Slam3dError) as e:
        capi.IcpHandle(p)
    assert e.value.code == -1
    # spec S4 (round 4): a wave's Gram sums must stay below 2^51 -- the farthest valid point below 90 m whatever the image size
    from slam3d_gx_amd import synth
    p = capi.default_params(synth.Intrinsics.scaled(64, 48))
    p.z_filter = 80.0                       # 64 x 48 x (80 m x 1.25)^2 < 2^28 passes the total-range test; r = 100 m does not pass the wave test
    with pytest.raises(capi.Slam3dError) as e:
        capi.IcpHandle(p)
    assert e.value.code == -1
    p = capi.default_params()
    p.coarse_iterations = -1
    with pytest.raises(capi.Slam3dError) as e:
        capi.IcpHandle(p)
    assert e.value.code == -1


def test_match_planes_is_exact_nearest_neighbour_without_gpu():
    """Row a9 (src/GraphicEnd.cpp:459-484): FLANN's approximate 4-d match on <= 3 planes, done exactly on the host."""
    import numpy as np
    rng = np.random.default_rng(0)
    a = rng.normal(size=(3, 4)).astype(np.float32)
    b = np.concatenate([a[[2, 0]] + np.float32(0.01), rng.normal(size=(2, 4)).astype(np.float32) + 5])
    idx, dist = capi.match_planes(a, b)
    d = np.linalg.norm(a[:, None, :].astype(np.float64) - b[None, :, :], axis=2)
    assert list(idx) == list(d.argmin(1)) and idx[0] == 1 and idx[2] == 0
    assert np.allclose(dist, d.min(1), rtol=1e-6)
    tie_idx, _ = capi.match_planes(a[:1], np.stack([a[0], a[0]]))
    assert list(tie_idx) == [0]                                    # ties -> lowest index
    none_idx, none_d = capi.match_planes(a, np.zeros((0, 4), np.float32))
    assert list(none_idx) == [-1, -1, -1] and np.isinf(none_d).all()

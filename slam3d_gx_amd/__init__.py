"""slam3d_gx_amd -- MI355X-native plane-ICP registration path for the slam3d_gx front end.

Only what the hot path needs lives here:
  csrc/   hand-written HIP kernels (gfx950) + the C-ABI implementation (include/slam3d_icp.h)
  host/   C++ host mirror of the reference's GraphicEnd pose API + run_SLAM-style driver
  capi.py ctypes binding of the C-ABI (plumbing for tests/bench; fails loudly if the .so is missing)
  synth.py deterministic synthetic frame pairs (BASELINE workloads)
"""
__version__ = "0.1.0"

"""Developer tool (MI355X): voxel grid, us per cloud at batch sizes 1 / 8 / 64, checked against the oracle."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from slam3d_gx_amd import capi, synth
import oracle_lib as O
pr = synth.make_pair(1000, 640, 480)
c = synth.backproject_numpy(pr.depth_src, pr.intr).reshape(-1, 4).copy()
c[:, 3] = np.random.default_rng(0).integers(0, 2 ** 32, c.shape[0], dtype=np.uint64).astype(np.uint32).view(np.float32)
dev = torch.device("cuda:0"); d = torch.from_numpy(c).to(dev)
h = capi.IcpHandle(capi.default_params(pr.intr, max_batch=1, device=0))
s = torch.cuda.Stream(device=dev)
from PIL import Image
dk = np.array(Image.open(os.path.join(ROOT, 'tests', 'golden', 'kinect', 'exp1_dep_1.png'))).astype(np.uint16)
lk = synth.backproject_numpy(dk, synth.Intrinsics(), z_filter=1e9).reshape(-1, 4)
lk = lk[np.isfinite(lk[:, 2])].copy(); lk[:, 3] = np.float32(0)            # the reference's PCD form: a raster-ordered LIST without the invalid pixels
for name, c in (("organized 640x480 (synthetic)", c), ("list of %d records (Kinect frame as convert2PCD writes it)" % len(lk), lk)):
  print(name)
  d = torch.from_numpy(c).to(dev)
  want = O.voxel_grid(c, 0.03, 7.0)
  for B in (1, 2, 4, 8, 64):
      ins = [d.clone() for _ in range(B)]; outs = [torch.zeros_like(d) for _ in range(B)]
      pin = [x.data_ptr() for x in ins]; pout = [x.data_ptr() for x in outs]; nn = [len(c)] * B
      for _ in range(3): ms = h.voxel_grid_batch_device(pin, nn, pout, 0.03, s.cuda_stream)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      R = 20
      for _ in range(R): ms = h.voxel_grid_batch_device(pin, nn, pout, 0.03, s.cuda_stream)
      torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
      ok = all(np.array_equal(o[:m].cpu().numpy().view(np.uint32), want.view(np.uint32)) for o, m in zip(outs, ms))
      print(f"B={B}: {1e6*dt/B:.2f} us per cloud, identical {ok}, HBM frac {(len(c)*16+ms[0]*16)/(dt/B)/8e12:.4f}")
print(h.voxel_grid_path_counts())

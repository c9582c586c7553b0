#!/usr/bin/env python3
"""Thread scaling of the CPU oracle (bench.py's cpu_baseline) on this host: 640x480 config-2 pair, 20 iterations, kd-tree NN.
usage: python tools/cpu_scaling.py [threads ...]      (ORC_TIMING=1 adds the phase times of every run on stderr)"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from slam3d_gx_amd import synth

pr = synth.make_pair(1000, 640, 480)
s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
counts = [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64, 128, os.cpu_count()]
print("OMP_PROC_BIND", os.environ.get("OMP_PROC_BIND"), "OMP_PLACES", os.environ.get("OMP_PLACES"), "cpus", os.cpu_count(),
      "affinity", len(os.sched_getaffinity(0)))
for th in counts:
    p = O.params(pr.intr, iterations=20, nn_method=1, threads=th)
    O.icp(s4, t4, p, trace=False)
    ts = []
    for _ in range(7 if th > 1 else 3):
        t0 = time.perf_counter(); O.icp(s4, t4, p, trace=False); ts.append(time.perf_counter() - t0)
    print(f"threads {th:4d}: {20 / statistics.median(ts):8.2f} it/s  (min {20 / max(ts):.1f} max {20 / min(ts):.1f})", flush=True)

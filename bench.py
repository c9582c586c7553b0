#!/usr/bin/env python3
"""bench.py -- ICP iterations/s of the MI355X plane-ICP path (BASELINE.json metric).

What is timed ("wall = first H2D enqueue -> last pose on host"):

  A *step* is one pass of the hot path over one batch of synthetic input: `--pairs-per-step` S frame pairs (default
  320: the driver's 20 steps then time 6,400 alignments, > 2 s), processed as S / P consecutive alignments of P pairs each (P = `--pairs`, default 1 = BASELINE config 2:
  a single 640x480 pair, 20 iterations, point-to-plane).  EVERY alignment inside the timed region consists of
      H2D of BOTH u16 depth images of the pair from pinned host memory (2 x 614 KB)
      -> back-projection -> target normals -> tile records of both frames -> 20 ICP iterations -> pose record on the host.
  Nothing is cached from one alignment to the next: the pairs cycle through a pool of `--pool` (default 16) DISTINCT
  seeded pairs per in-flight handle, so the ownership map, the hints and the caches are never primed by the identical
  frame.  `--in-flight` handles (one HIP stream each) take turns, i.e. the stream of independent pairs is software
  pipelined; `single_step_latency_ms` is the same work with one alignment at a time.
  The synthetic frames are BASELINE.md section 4's workload as specified (depth noise sigma = 0.0012 z^2, 8x8-pixel Bernoulli
  holes at p = 0.25; round 5 -- rounds 1-4 quoted a low-noise surrogate, sigma = 0.0002 z^2 with 32x32 holes at p = 0.2, which is
  now the leg `low_noise_surrogate`).  `config.coarse_iterations` of every run's `iterations` take a quarter of the sources
  (spec S4c); the line says so and the roofline's algorithmic bytes count them at a quarter.
  The former headline (one pair resident in HBM, re-run in place, preprocessing included) is `resident_same_pair_value`;
  `plane_normals` is the same stream under SLAM3D_EST_PLANE (the frames' planes give the normals) on the headline pairs and on
  the reference's Kinect frames.

  N > 1: every rank processes its own pairs (seeds offset by rank, no data-path collective) and the S pose records of
  a step are all-gathered over RCCL *inside the library* (slam3d_pose_gather_*), overlapping the next step (weak
  scaling).  `--mode dense` is BASELINE config 5: ONE pair whose source rows are sharded over the ranks with one
  all-reduce of the iteration's integer Gram totals (16 x 40 int64 in place; 36 int64 in the three-step form) per iteration
  (slam3d_icp_dense_run; strong scaling).

Prints ONE JSON line on rank 0.  Beside metric/value/...:
  roofline            dominant kernel k_nn_tiles_acc accounted against HBM: algorithmic bytes per launch (SURVEY.md
                      8(d): 12 B xyz + 4 B idx per valid source point, 12 B xyz + 12 B normal per valid target) / mean
                      launch duration, measured with HIP events on the launch stream over `--profile-aligns` alignments.
  roofline_bruteforce the north-star algorithm (full brute-force scan) on the matrix cores (bf16-split and f32 forms) and on the VALU.
  cpu_baseline        the CPU oracle timed on the host on a bounded sample of the same workload.
  config3 / config5   BASELINE configs 3 (64 pairs per launch) and 5 (1280x960 dense) measured in the same run, each
                      with its own roofline and cpu_baseline (N = 1 only; skip with --no-extra-configs).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md:40-41 (vector == f32-MFMA dense peak)
HBM_PEAK_GBPS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md:35 (spec; ~6.3 TB/s achievable)

_FINAL = []      # rank 0's result line, printed after all teardown so that it is the last line on stdout


LINE_LIMIT = 6000    # bytes of the final stdout line (round 5's 21.5 KB line did not parse on the driver's side: VERDICT r5 item 1)
_LEGS_FILE = [os.path.join(ROOT, "gpurun_out", "bench_legs.json")]


def _rnd(x, sig=6):
    """floats to `sig` significant digits, recursively (the full-precision values are in the legs file)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _rnd(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_rnd(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "algorithmic_bytes_per_launch", "valu_issue_frac")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "min_value", "median_value", "spread", "numa_local", "single_thread_value", "bruteforce_value")


def _roof(r):
    if not isinstance(r, dict):
        return None
    o = _pick(r, _ROOF_KEYS)
    o.setdefault("traffic", None)
    if "kernel" in o:
        o["kernel"] = o["kernel"].split(" (")[0]
    return o


def _cpu(c):
    if not isinstance(c, dict):
        return None
    o = _pick(c, _CPU_KEYS)
    if "sample" in o:
        o["sample"] = _short(o["sample"], 150)
    return o


def _leg_numbers(v):
    """a side leg as its few numbers (value + what the review quotes); nested legs one level down"""
    if not isinstance(v, dict):
        return v
    keys = ("value", "ratio_to_headline", "single_step_latency_ms", "nn_launch_us", "preprocess_us", "icp_only_value", "us_per_iteration",
            "us_per_frame", "single_call_us", "status", "residual_rot_rad", "residual_trans_m", "cpu_value", "vs_cpu")
    o = _pick(v, keys)
    for k, w in v.items():
        if isinstance(w, dict) and ("value" in w or "icp_only_value" in w or "us_per_frame" in w) and k not in ("roofline", "cpu_baseline"):
            o[k] = _pick(w, keys)
            if isinstance(w.get("cpu_baseline"), dict):
                o[k]["cpu_value"] = w["cpu_baseline"].get("value")
    if isinstance(v.get("cpu_baseline"), dict):
        o["cpu_value"] = v["cpu_baseline"].get("value")
    return o


_LEG_NAMES = ("all_sources_every_iteration", "low_noise_surrogate", "plane_normals", "real_pair", "voxel_icp", "two_pairs_per_launch")


def compact_line(out, legs_file=None, limit=LINE_LIMIT):
    """The ONE line the driver parses: the bench contract's fields, `roofline`, `cpu_baseline`, parity, `survey_8d` and compact
    `config3` / `config5` -- nothing else.  Every leg, note, per-iteration array and thread curve is in `legs_file` (the complete
    object this line is cut from).  Guaranteed shorter than `limit` bytes: optional parts are dropped, last first, until it is."""
    c = out.get("config", {})
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = out.get("vs_baseline")
    line.update(_pick(out, ("dtype", "data")))
    cfg = _pick(c, ("workload", "pairs_per_step_per_gpu", "pairs_per_launch", "alignments_per_step", "iterations", "estimator", "coarse_iterations",
                    "synthetic_workload", "noise_sigma_over_z2", "hole_block_px", "hole_prob", "h2d_bytes_per_pair", "nn_mode", "n_src", "n_tgt",
                    "parallelism", "gathered_pose_records", "gathered_seeds", "in_flight"))
    for k, n in (("workload", 420), ("parallelism", 160), ("synthetic_workload", 60)):
        if k in cfg:
            cfg[k] = _short(cfg[k], n)
    line["config"] = cfg
    line["roofline"] = _roof(out.get("roofline"))
    line["cpu_baseline"] = _cpu(out.get("cpu_baseline"))
    line.update(_pick(out, ("parity_vs_oracle", "survey_8d", "single_step_latency_ms", "timed_region_s", "status", "rccl_ranks", "pose_exchange",
                            "kernel_only_value", "predicted_scaling")))
    if "pose_exchange" in line:
        line["pose_exchange"] = _short(line["pose_exchange"], 80)
    if "predicted_scaling" in line:
        line["predicted_scaling"] = _pick(line["predicted_scaling"], ("vs_1_gpu",))
    optional = []
    for name in ("config3", "config5"):
        v = out.get(name)
        if isinstance(v, dict):
            o = _pick(v, ("value", "unit", "ms_per_alignment", "ms_per_step", "kernel_only_value", "n_src", "n_tgt", "status_ok", "parity_vs_oracle"))
            o["workload"] = _short(v.get("workload", ""), 130)
            o["roofline"] = _roof(v.get("roofline"))
            o["cpu_baseline"] = _pick(v.get("cpu_baseline") or {}, ("value", "unit", "cores", "kind", "min_value", "median_value"))
            line[name] = o
            optional.append(name)
    legs = {k: _leg_numbers(out[k]) for k in _LEG_NAMES if k in out}
    if legs:
        line["legs"] = legs
    if legs_file:
        line["legs_file"] = os.path.relpath(legs_file, ROOT)
    line = _rnd(line)
    # shrink until it fits: side-leg numbers first, then per-rank-free extras, then the secondary configs' cpu baselines
    for drop in (("legs",), ("predicted_scaling",), ("kernel_only_value",), ("config5", "cpu_baseline"), ("config3", "cpu_baseline"),
                 ("config5",), ("config3",), ("survey_8d",)):
        s = json.dumps(line, separators=(",", ":"))
        if len(s) < limit:
            return s
        tgt = line
        for k in drop[:-1]:
            tgt = tgt.get(k, {}) if isinstance(tgt, dict) else {}
        if isinstance(tgt, dict):
            tgt.pop(drop[-1], None)
    s = json.dumps(line, separators=(",", ":"))
    if len(s) >= limit:
        line["config"] = _pick(line["config"], ("workload", "iterations", "estimator", "coarse_iterations"))
        line["config"]["workload"] = _short(line["config"].get("workload", ""), 160)
        s = json.dumps(line, separators=(",", ":"))
    return s


def _finish(out):
    """rank 0: keep the complete object in the legs file (and, one short line per leg, on stdout BEFORE the final line) and queue
    the compact line"""
    path = _LEGS_FILE[0]
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f)
    except OSError as e:
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
        path = None
    for k in _LEG_NAMES + ("overlap", "roofline_bruteforce", "per_rank"):
        if k in out:
            v = out[k]
            if k == "overlap":
                v = _pick(v, ("value_while_stamping", "mean_resident_nn_kernels", "nn_launch_us_overlapped", "alignment_latency_us", "nn_to_nn_gap_us"))
            elif k == "roofline_bruteforce":
                v = dict(_roof(v), **_pick(v, ("equivalent_f32_contraction_tflops", "equivalent_frac_of_f32_peak", "iterations_per_s_if_used")))
            elif k != "per_rank":
                v = _leg_numbers(v)
            print(json.dumps({"leg": k, **({"v": _rnd(v)} if not isinstance(v, dict) else _rnd(v))}, separators=(",", ":")), flush=True)
    whole = json.dumps(out)
    _FINAL.append(whole if len(whole.encode()) < LINE_LIMIT else compact_line(out, path))


def _emit_final():
    """RCCL prints a version banner through C stdio; flush that first, then print the one JSON line and flush."""
    if not _FINAL:
        return
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(_FINAL[-1], flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=1, help="frame pairs per launch sequence and GPU (config 3: 64)")
    ap.add_argument("--pairs-per-step", type=int, default=480, help="frame pairs one step processes per GPU (S / --pairs alignments)")
    ap.add_argument("--pool", type=int, default=16, help="distinct seeded pairs per in-flight handle that the stream cycles through")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--estimator", choices=["point2plane", "svd", "plane", "plane_gate"], default="point2plane",
                    help="plane: SLAM3D_EST_PLANE (per-plane normals); plane_gate: the same with the plane-pair gate")
    ap.add_argument("--coarse-iterations", type=int, default=3, help="spec S4c: leading iterations of a run on a quarter of the sources (0: none)")
    ap.add_argument("--nn-mode", type=int, default=0, help="0 auto(tiles) 1 brute-force VALU 2 brute-force MFMA 3 tiles")
    ap.add_argument("--mode", choices=["batch", "dense", "seg", "voxel"], default="batch",
                    help="seg: row f-2, batched RANSAC plane segmentation of --pairs frames per GPU; "
                         "voxel: row f-1, PassThrough + VoxelGrid(0.03) of one resident cloud per step")
    ap.add_argument("--noise-sigma", type=float, default=0.0012, help="depth noise sigma/z^2 of the synthetic frames (BASELINE.md section 4: 0.0012; rounds 1-4: 0.0002)")
    ap.add_argument("--hole-block", type=int, default=8, help="edge of the invalid-pixel blocks at 640x480 (BASELINE.md section 4: 8; rounds 1-4: 32)")
    ap.add_argument("--hole-prob", type=float, default=0.25, help="probability of an invalid block (BASELINE.md section 4: 0.25; rounds 1-4: 0.2)")
    ap.add_argument("--overlap-aligns", type=int, default=256, help="alignments of the stamped pass that measures how many NN launches are resident at once")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bruteforce", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the config-3 / config-5 / survey-noise / resident legs")
    ap.add_argument("--timed-only", action="store_true", help="stop after the timed region (kernel traces of the pipelined regime alone)")
    ap.add_argument("--profile-aligns", type=int, default=48, help="alignments of the event-profiled pass (roofline launch time)")
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--legs-file", default="", help="where rank 0 writes the COMPLETE result object (every leg, note and per-iteration array); "
                    "default gpurun_out/bench_legs.json.  The final stdout line carries the contract fields only and stays under 6 KB")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + --one-device: exercise the N>1 code path with several ranks on ONE GPU (tests only)")
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--force-collective", action="store_true", help="developer knob: run the RCCL exchanges even with one rank")
    ap.add_argument("--no-pipeline", action="store_true", help="one handle, every alignment fetched before the next is queued")
    ap.add_argument("--dump-table", default="", help="rank 0 writes the LAST step's gathered pose table (pair order) with every record's seed "
                    "to this JSON file (tests: BASELINE config 4's shape, --gpus 8 --pairs 64 --pairs-per-step 64 --in-flight 1 --pool 64)")
    ap.add_argument("--in-flight", type=int, default=8, help="alignments in flight (handles taking turns, one HIP stream each; the runtime has "
                    "four hardware queues by default, so multiples of four: 4 -> 84 k, 8 -> 86 k, 12 / 16 the same; 5 -> 63 k, 6 -> 73 k it/s)")
    return ap.parse_args()


class Est:
    """--estimator as the library's and the oracle's parameters"""

    def __init__(self, capi, name, coarse=3):
        self.name = name
        self.lib, self.flags, self.orc, self.orc_gate = {
            "point2plane": (capi.EST_POINT2PLANE, 0, 0, 0), "svd": (capi.EST_SVD, 0, 1, 0),
            "plane": (capi.EST_PLANE, 0, 2, 0), "plane_gate": (capi.EST_PLANE, capi.PLANE_PAIR_GATE, 2, 1)}[name]
        self.normals = self.lib != capi.EST_SVD          # the target carries normals (24 B per target point instead of 12)
        self.coarse = coarse

    def kw(self):
        return dict(estimator=self.lib, plane_flags=self.flags, coarse_iterations=self.coarse)

    def okw(self):
        return dict(estimator=self.orc, plane_pair_gate=self.orc_gate, coarse_iterations=self.coarse)


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the driver compares it)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "ICP iterations/sec on 640×480 clouds; SE(3) pose error vs PCL ref"


def kernel_src_sha16():
    h = hashlib.sha256()
    with open(os.path.join(ROOT, "slam3d_gx_amd", "csrc", "icp_kernels.hpp"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def committed_traffic(tag):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile of THIS kernel source (the profile
    records the SHA-256 of icp_kernels.hpp it was taken on; a stale profile yields null, never a stale number).  The
    newest profiles/rNN_traffic.json whose hash matches is used."""
    import glob
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            e = d[tag]
        except Exception:
            continue
        if d.get("kernel_src_sha16") == kernel_src_sha16():
            return e["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), e
        stale = stale or os.path.relpath(path, ROOT) + " (stale: kernel source changed since the PMC pass)"
    return None, stale, {}


# ------------------------------------------------------------------------------------------------ workload
class Pool:
    """`n` distinct seeded frame pairs as u16 depth images in PINNED host memory (so that hipMemcpyAsync is truly
    asynchronous), plus their float4 clouds for the oracle legs."""

    CACHE = {}          # (seed, width, height, sigma) -> FramePair, filled in parallel by Pool.prefetch

    @classmethod
    def prefetch(cls, synth, specs):
        todo = [sp for sp in dict.fromkeys(specs) if sp not in cls.CACHE]
        if todo:
            cls.CACHE.update(synth.make_pairs(todo))

    @staticmethod
    def spec(seed, width, height, sigma, mask=None):
        """mask = (hole_block, hole_prob) or None for the default 32 / 0.2"""
        return (seed, width, height, sigma) + (tuple(mask) if mask and tuple(mask) != (32, 0.2) else ())

    def __init__(self, torch, synth, seeds, width, height, sigma, mask=None, pairs=None):
        if pairs is None:
            self.prefetch(synth, [self.spec(s, width, height, sigma, mask) for s in seeds])
            pairs = [self.CACHE[self.spec(s, width, height, sigma, mask)] for s in seeds]
        self.pairs = list(pairs)
        self.intr = self.pairs[0].intr
        self.depth = torch.empty((len(self.pairs), 2, height, width), dtype=torch.int16).pin_memory()     # u16 bits
        arr = self.depth.numpy().view(np.uint16)
        for i, p in enumerate(self.pairs):
            arr[i, 0] = p.depth_src
            arr[i, 1] = p.depth_tgt
        self.stride = 2 * height * width * 2
        self.frame_bytes = height * width * 2
        self.base = self.depth.data_ptr()

    def src_ptr(self, i):
        return self.base + i * self.stride

    def tgt_ptr(self, i):
        return self.base + i * self.stride + self.frame_bytes

    def __len__(self):
        return len(self.pairs)


class Streamer:
    """Software-pipelined stream of alignments over `handles`: alignment k uploads both depth images of its P pairs,
    runs, and its poses are fetched after the following len(handles)-1 alignments were queued."""

    def __init__(self, handles, pools, P, on_fetch=None, T_init=None):
        self.handles, self.pools, self.P = handles, pools, P
        self.on_fetch = on_fetch    # called as on_fetch(handle_index, handle) after every fetch (the stamped pass reads the launch stamps there)
        self.T_init = T_init        # optional initial guess of every alignment (the real-frame leg)
        self.k = 0
        self.queue = []             # handles with a run in flight, oldest first
        self.seedq = []             # the seeds of their pairs, same order
        self.last = None
        self.last_seeds = None
        self.t_enqueue = self.t_fetch = 0.0      # host seconds spent queueing alignments / waiting for their poses (per-rank attribution, N > 1)
        self.n_enqueued = 0

    def _enqueue(self, hi):
        t0 = time.perf_counter()
        h, pool = self.handles[hi], self.pools[hi]
        n = len(pool)
        base = (self.k // len(self.handles)) * self.P
        seeds = []
        for i in range(self.P):
            j = (base + i) % n
            seeds.append(pool.pairs[j].seed)
            h.frame_set_depth_host_ptr(2 * i, pool.src_ptr(j))
            h.frame_set_depth_host_ptr(2 * i + 1, pool.tgt_ptr(j))
            h.set_pair(i, 2 * i, 2 * i + 1)
        h.run(self.P, self.T_init)
        self.queue.append(hi)
        self.seedq.append(seeds)
        self.k += 1
        self.t_enqueue += time.perf_counter() - t0
        self.n_enqueued += 1

    def run(self, n_align, sink=None):
        """n_align alignments; every one's poses reach the host (and `sink`) before this returns"""
        nh = len(self.handles)
        for _ in range(n_align):
            if len(self.queue) >= nh:
                self._drain_one(sink)
            self._enqueue(self.k % nh)
        while self.queue:
            self._drain_one(sink)
        return self.last

    def _drain_one(self, sink):
        hi = self.queue.pop(0)
        t0 = time.perf_counter()
        self.last = self.handles[hi].fetch_results(self.P)
        self.t_fetch += time.perf_counter() - t0
        self.last_seeds = self.seedq.pop(0)
        if self.on_fetch is not None:
            self.on_fetch(hi, self.handles[hi])
        if sink is not None:
            sink.extend(self.last)
            if hasattr(sink, "seeds"):
                sink.seeds.extend(self.last_seeds)


def cpu_thread_curve(s4, t4, width, height, okw, iterations, counts, runs, T_init=None, single_iterations=None):
    """The CPU oracle timed at every thread count of `counts`, each in a process of its own (bench_cpu_worker.py): the team pinned to
    `count` physical cores (one NUMA node first, then its neighbours) before libgomp starts, a warm-up run, then `runs` timed alignments.
    Returns {count: {value (median), min_value, max_value, spread (IQR / median), range, iterations, runs, numa_local}}, and the count to
    quote: the fastest median among the points whose spread is <= 20 % (round 5 quoted a point
    whose samples differed by 100 %), or -- if none is that steady -- the steadiest point."""
    import subprocess
    import tempfile
    curve = {}
    with tempfile.TemporaryDirectory() as tmp:
        npz = os.path.join(tmp, "pair.npz")
        np.savez(npz, s4=np.ascontiguousarray(s4, dtype=np.float32), t4=np.ascontiguousarray(t4, dtype=np.float32))
        for th in counts:
            its = iterations if (th > 1 or single_iterations is None) else min(single_iterations, iterations)
            # from 16 threads on a team is timed twice: packed on one NUMA node ("close") and dealt over the nodes ("spread" -- the search
            # is bound by memory latency, a second socket's caches can beat locality); the curve keeps the faster placement
            for placement in (("close", "spread") if th >= 16 else ("close",)):
                spec = {"width": int(width), "height": int(height), "threads": int(th), "runs": int(runs if th > 1 else min(runs, 3)), "placement": placement,
                        "params": dict(okw, iterations=int(its), nn_method=1), "T_init": None if T_init is None else np.asarray(T_init, dtype=np.float64).reshape(16).tolist()}
                sp = os.path.join(tmp, f"spec{th}{placement}.json")
                with open(sp, "w") as f:
                    json.dump(spec, f)
                env = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
                try:
                    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench_cpu_worker.py"), npz, sp], capture_output=True, text=True, timeout=600, env=env)
                    d = json.loads(pr.stdout.strip().splitlines()[-1])
                except Exception as e:      # noqa: BLE001 -- a thread count that cannot be timed is left out of the curve
                    print(f"bench.py: cpu baseline worker failed at {th} threads ({placement}): {e}", file=sys.stderr)
                    continue
                t = sorted(d["times_s"])
                med = statistics.median(t)
                # spread = interquartile range / median (one preempted run out of seven must not disqualify a point); range = (max - min) / median
                q1, q3 = t[len(t) // 4], t[(3 * len(t)) // 4 if len(t) >= 4 else -1]
                pt = {"value": its / med, "min_value": its / t[-1], "max_value": its / t[0], "spread": (q3 - q1) / med, "range": (t[-1] - t[0]) / med,
                      "iterations": its, "runs": len(t), "numa_local": d["numa_local"], "placement": placement}
                if th not in curve or (pt["spread"] <= 0.20 and (curve[th]["spread"] > 0.20 or pt["value"] > curve[th]["value"])):
                    curve[th] = pt
    multi = [c for c in curve if c > 1] or list(curve)
    steady = [c for c in multi if curve[c]["spread"] <= 0.20]
    best = max(steady, key=lambda c: curve[c]["value"]) if steady else min(multi, key=lambda c: curve[c]["spread"])
    return curve, best


def cpu_baseline_leg(pair, s4, t4, iterations, est, gpu_result, gpu_idx, brute_sample=True):
    """Times the CPU oracle on the same pair and checks the GPU result against it.
    (oracle use is confined to this leg: checker + CPU baseline, never the measured path)

    `value` is the median of 7 alignments at the thread count that is fastest AMONG THE STEADY ONES (spread <= 20 %) of a
    1 / 8 / 16 / 32 / 64 / physical cores curve; every point runs in its own process with the OpenMP team pinned to physical cores
    of one NUMA node (cpu_thread_curve).  Measured on the GPU host (2 x EPYC 9575F, 256 hardware threads): the 217 k queries of a
    pair do not feed more than a few dozen threads (HISTORY.md section 9); `min_value` / `median_value` / `spread` of the quoted point
    and the whole curve with its spreads are in the object."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = os.cpu_count() or 1
    big = pair.intr.width * pair.intr.height > 640 * 480
    counts = sorted({c for c in ((1, 16, 64) if big else (1, 8, 16, 32, 64, cores // 2)) if 1 <= c <= cores})
    curve, best = cpu_thread_curve(s4, t4, pair.intr.width, pair.intr.height, est.okw(), iterations, counts, 5 if big else 7,
                                   single_iterations=4 if big else 10)
    # the checker's copy of the result (any thread count gives the same bits)
    ro = O.icp(s4, t4, O.params(pair.intr, iterations=iterations, nn_method=1, threads=min(16, cores), **est.okw()), trace=True)
    phases = None
    if not big:      # where one run at the best thread count spends its time (the oracle prints its phase times on request)
        try:
            import subprocess
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                np.savez(os.path.join(tmp, "pair.npz"), s4=s4, t4=t4)
                code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import oracle_lib as O; from slam3d_gx_amd import synth; "
                        "z = np.load(%r); intr = synth.Intrinsics.scaled(%d, %d); "
                        "p = O.params(intr, estimator=%d, plane_pair_gate=%d, coarse_iterations=%d, iterations=%d, nn_method=1, threads=%d); O.icp(z['s4'], z['t4'], p); O.icp(z['s4'], z['t4'], p)"
                        % (os.path.join(ROOT, "tests"), ROOT, os.path.join(tmp, "pair.npz"), pair.intr.width, pair.intr.height, est.orc, est.orc_gate, est.coarse, iterations, best))
                pr_ = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, ORC_TIMING="1"))
            ln = [l for l in pr_.stderr.splitlines() if l.startswith("orc_icp timing")]
            if ln:
                phases = ln[-1]
        except Exception:       # noqa: BLE001 -- informational only
            pass
    out = {
        "value": curve[best]["value"], "unit": "ICP iterations/s", "cores": best, "kind": "port", "host_hardware_threads": cores,
        "median_value": curve[best]["value"], "min_value": curve[best]["min_value"], "spread": curve[best]["spread"], "numa_local": curve[best]["numa_local"],
        "placement": curve[best]["placement"],
        "sample": f"oracle kd-tree ICP, {best} threads pinned to physical cores ({curve[best]['placement']}: the faster of one-node-first / dealt over the nodes; own process), 1 pair seed {pair.seed} x {iterations} "
                  f"iterations incl. normals + kd-tree build, median of {curve[best]['runs']} after a warm-up; fastest thread count with spread <= 20 %",
        "thread_curve": {str(k): round(v["value"], 2) for k, v in curve.items()},
        "thread_curve_spread": {str(k): round(v["spread"], 3) for k, v in curve.items()},
        "single_thread_value": curve[1]["value"] if 1 in curve else None,
        "single_thread_sample": f"same, 1 thread, {curve[1]['iterations'] if 1 in curve else 0} iterations (PCL's own ICP is single-threaded)",
        "phase_ms_one_run": phases,
    }
    if brute_sample:
        # cpu_A of SURVEY.md 8(d): the literal brute-force scan with the canonical arithmetic, all cores, ONE NN pass
        p_br = O.params(pair.intr, iterations=1, nn_method=0, threads=0, **est.okw())
        t0 = time.perf_counter()
        O.nn_once(s4, t4, p_br, T=None, use_normals={0: 1, 1: 0, 2: 2}[est.orc])
        t_br = time.perf_counter() - t0
        out["bruteforce_value"] = 1.0 / t_br
        out["bruteforce_sample"] = f"cpu_A: literal brute-force NN scan (canonical fp32 arithmetic), {min(cores, 32)} threads (the oracle's default team), one pass"
    parity = None
    if gpu_result is not None:
        rot, tr = O.pose_error(ro["T_trace"][-1], gpu_result["T_raw"])
        parity = {
            "rot_err_rad": rot, "trans_err_m": tr,
            "idx_mismatches": int((gpu_idx != ro["idx"]).sum()) if gpu_idx is not None else None,
            "T_bit_identical": bool(np.array_equal(ro["T_trace"][-1], gpu_result["T_raw"])),
        }
    return out, parity


BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak of one MI355X (MI355X_MICROARCH.md; 2:1-sparsity figures are never used)


def bruteforce_leg(capi, intr, est, s4, t4, local_rank, iterations=4):
    """The north-star algorithm on one pair of the pool: every source x every target, the distance step as a dense
    contraction on the matrix cores.  Three kernels, same bits: k_nn_mfma16 (each float split into three bf16 terms, one
    v_mfma_f32_16x16x32_bf16 per 16 x 16 tile -- the default of NN_BRUTE_MFMA), k_nn_mfma (v_mfma_f32_16x16x4_f32;
    SLAM3D_MFMA_BF16=0) and the fp32-VALU scan k_nn_valu."""
    out = {}
    for mode, name, env in ((capi.NN_BRUTE_MFMA, "mfma16", None), (capi.NN_BRUTE_MFMA, "mfma", ("SLAM3D_MFMA_BF16", "0")), (capi.NN_BRUTE_VALU, "valu", None),
                            (capi.NN_BRUTE_VALU, "valu_filter", ("SLAM3D_VALU_FILTER", "1"))):
        # (coarse_iterations = 0: every launch scans every source against every target -- the N x M contraction the flop count assumes)
        params = capi.default_params(intr, iterations=iterations, max_batch=1, device=local_rank, nn_mode=mode, **dict(est.kw(), coarse_iterations=0))
        if env:
            os.environ[env[0]] = env[1]
        try:
            with capi.IcpHandle(params) as h:
                h.set_clouds_host(0, s4, t4)
                h.set_profiling(True)
                h.run(1)
                h.fetch_results(1)                       # warm-up
                h.set_clouds_host(0, s4, t4)
                h.run(1)
                r = h.fetch_results(1)[0]
                ms = float(np.mean(h.get_iteration_timings()[1:]))    # iteration 0 has no previous match to bound the filter
        finally:
            if env:
                del os.environ[env[0]]
        out[name] = (ms, float(r["n_src"]) * float(r["n_tgt"]))
    ms16, pairs = out["mfma16"]
    ms, _ = out["mfma"]
    vms, _ = out["valu"]
    vfms, _ = out["valu_filter"]
    # executed flops: the bf16 instruction multiplies K = 32 slots per (source, target) pair (64 flop; 22 slots carry terms),
    # the f32 one K = 4 (8 flop), the VALU scan evaluates the canonical distance (8 flop)
    ach16 = 64.0 * pairs / (ms16 * 1e-3) / 1e12
    ach = 8.0 * pairs / (ms * 1e-3) / 1e12
    return {"kernel": "k_nn_mfma16 (full brute-force scan: |q|^2 - 2 p.q - thr as ONE v_mfma_f32_16x16x32_bf16 per 16x16 tile, every float split "
                      "exactly into three bf16 terms; conservative filter + exact fp32 re-evaluation of flagged pairs; bit-identical results)",
            "bound": "mfma", "achieved": ach16, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach16 / BF16_PEAK_TFLOPS,
            "traffic": None, "launch_ms": ms16, "flops_per_launch": 64.0 * pairs, "iterations_per_s_if_used": 1e3 / ms16,
            "equivalent_f32_contraction_tflops": 8.0 * pairs / (ms16 * 1e-3) / 1e12,
            "equivalent_frac_of_f32_peak": 8.0 * pairs / (ms16 * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
            "note": "achieved counts the flops the bf16 instruction executes (64 per pair) against the dense bf16 peak; the same scan as an f32 "
                    "K = 4 contraction is 8 flop per pair: equivalent_f32_contraction_tflops, above the 157.3 TF f32 matrix peak",
            "f32_mfma_kernel": {"kernel": "k_nn_mfma (v_mfma_f32_16x16x4_f32, SLAM3D_MFMA_BF16=0)", "launch_ms": ms, "achieved": ach,
                                "peak": FP32_PEAK_TFLOPS, "frac": ach / FP32_PEAK_TFLOPS},
            "valu_kernel": {"kernel": "k_nn_valu (same scan on the fp32 VALU, canonical distances)", "launch_ms": vms,
                            "achieved": 8.0 * pairs / (vms * 1e-3) / 1e12, "frac": 8.0 * pairs / (vms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS},
            "valu_filter_kernel": {"kernel": "k_nn_valu<filter> (SLAM3D_VALU_FILTER=1: the expanded form |q|^2 - 2 p.q as a filter on the VALU, 6 flop per pair executed; "
                                             "flagged chunks re-evaluated canonically)", "launch_ms": vfms,
                                   "achieved": 6.0 * pairs / (vfms * 1e-3) / 1e12, "frac": 6.0 * pairs / (vfms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                   "equivalent_f32_contraction_tflops": 8.0 * pairs / (vfms * 1e-3) / 1e12}}


def alg_bytes_per_launch(n_src, n_tgt, est, iterations):
    """12 B xyz + 4 B idx per valid source point, 12 B xyz (+ 12 B normal) per valid target point, each array once
    per iteration -- as the MEAN over a run's launches: the first min(coarse, iterations - 1) launches take the sources of every
    fourth 8x8-pixel tile (spec S4c), counted at n_src / 4 (the round-4 line counted them in full)."""
    nc = max(0, min(est.coarse, iterations - 1))
    src_share = 1.0 - 0.75 * nc / max(iterations, 1)
    return 16.0 * n_src * src_share + (24.0 if est.normals else 12.0) * n_tgt


def profiled_pass(handle, pool, P, n_align, est):
    """Event-profiled alignments on ONE handle, one at a time (every event record serialises the stream, so this is
    kept out of the timed region): mean NN launch duration, kernel-only time per alignment, algorithmic bytes."""
    handle.set_profiling(True)
    nn, tot, pre, alg, flops, per_it = [], [], [], [], [], None
    st = Streamer([handle], [pool], P)
    for _ in range(n_align):
        res = st.run(1)
        tm = handle.get_timings()
        nn.append(tm["nn_ms"]); tot.append(tm["total_ms"]); pre.append(tm["preprocess_ms"])
        alg.append(sum(alg_bytes_per_launch(r["n_src"], r["n_tgt"], est, handle.params.iterations) for r in res))
        flops.append(sum(8.0 * r["n_src"] * r["n_tgt"] for r in res))
        per_it = handle.get_iteration_timings()
    handle.set_profiling(False)
    return dict(nn_ms=statistics.mean(nn), total_ms=statistics.mean(tot), preprocess_ms=statistics.mean(pre),
                alg_bytes=statistics.mean(alg), flops=statistics.mean(flops), per_it=[round(float(x), 4) for x in per_it])


def tiles_roofline(prof, iterations, tag, note_extra=""):
    launch_ms = prof["nn_ms"] / max(iterations, 1)
    ach = prof["alg_bytes"] / (launch_ms * 1e-3) / 1e9
    traffic, src, pm = committed_traffic(tag)
    floor = pm.get("valu_issue_floor_us")
    return {
        "kernel": "k_nn_tiles_acc (exact tile-pruned NN + fused normal-equation accumulation)",
        "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
        "traffic": traffic, "traffic_source": src, "launch_ms": launch_ms,
        "launch_ms_source": "HIP events around every k_nn_tiles_acc launch on the launch stream, separate pass (alignments one at a time)",
        "algorithmic_bytes_per_launch": prof["alg_bytes"],
        "valu_issue_floor_us": floor, "valu_instructions_per_wave": pm.get("valu_instructions_per_wave"),
        "valu_issue_frac": (floor / (launch_ms * 1e3)) if floor else None,
        "equivalent_bruteforce_tflops": prof["flops"] / (launch_ms * 1e-3) / 1e12,
        "note": ("streaming accounting (each array once per iteration); the kernel is VALU-issue/latency bound, not HBM bound "
                 "(DESIGN.md section 6); equivalent_bruteforce_tflops = flops of a full scan / this launch time" + note_extra),
    }


def overlap_pass(torch, handles, pools, P, n_align, iterations):
    """How many NN launches of the in-flight alignments are really resident at once -- measured WITHOUT a tracer.  Every
    handle stamps its launches (first block's start, last wave's end) with the GPU's constant-rate 100 MHz real-time
    counter, common to all streams (slam3d_icp_set_stamping: fire-and-forget atomics, no stream serialisation; a kernel
    tracer serialises the four streams and halves the overlap it is supposed to show).  The same pipelined stream as the
    timed region runs for n_align alignments; the stamps of every alignment are read at its fetch."""
    nh = len(handles)
    per_handle = (n_align + nh - 1) // nh
    n_align = per_handle * nh
    st = Streamer(handles, pools, P)
    for h in handles:
        h.set_stamping(per_handle)            # the ring holds exactly the measured runs of this handle: nothing is copied meanwhile
    st.run(2 * nh)                            # re-capture the graphs with the stamp ring, fill the pipeline (these runs fall out of the ring)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.run(n_align)
    torch.cuda.synchronize()
    host_s = time.perf_counter() - t0
    per = [h.get_stamps().astype(np.int64) for h in handles]          # [runs, 2 iterations, 2] per handle, oldest first
    for h in handles:
        h.set_stamping(0)
    # alignment k of the stream ran on handle k % nh as that handle's run k // nh
    recs = [(k % nh, per[k % nh][k // nh]) for k in range(n_align) if k // nh < len(per[k % nh])]
    trim = 2 * nh
    body = recs[trim:len(recs) - trim] if len(recs) > 4 * trim else recs
    TICK = 1e-8
    nn = np.array([r[1][:iterations] for r in body])                  # [align][it][2]
    sv = np.array([r[1][iterations:2 * iterations] for r in body])
    ok_sv = (sv[..., 1] > 0) & (sv[..., 0] < (1 << 62))
    a_start = nn[:, 0, 0]
    a_end = np.maximum(nn[:, -1, 1], np.where(ok_sv[:, -1], sv[:, -1, 1], 0))
    lo, hi = int(a_start.min()), int(a_end.max())
    iv = nn.reshape(-1, 2)
    dur = (iv[:, 1] - iv[:, 0]).astype(np.float64)
    # sweep line: time spent with k NN launches resident
    ev = np.concatenate([np.stack([iv[:, 0], np.ones(len(iv), np.int64)], 1), np.stack([iv[:, 1], -np.ones(len(iv), np.int64)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    level_t = {}
    cur, last = 0, lo
    for t, d in ev:
        if t > last:
            level_t[cur] = level_t.get(cur, 0) + (t - last)
            last = t
        cur += d
    window = float(hi - lo)
    busy = window - level_t.get(0, 0)
    n_al = len(body)
    per_align_wall_us = window * TICK * 1e6 / n_al               # device clock: window / alignments completed in it
    sum_nn_per_align_us = dur.sum() * TICK * 1e6 / n_al
    conc_window = dur.sum() / window
    out = {
        "method": ("launch stamps: wall_clock64 (100 MHz real-time counter, common to all XCDs and streams) folded by every block of "
                   "every NN launch into (min start, max end) with fire-and-forget atomics; no tracer, no HIP events, all handles live"),
        "alignments_analysed": n_al, "in_flight": nh, "pairs_per_alignment": P,
        "value_while_stamping": n_align * P * iterations / host_s,
        "nn_launch_us_overlapped": {"mean": float(dur.mean() * TICK * 1e6), "p50": float(np.percentile(dur, 50) * TICK * 1e6),
                                    "p90": float(np.percentile(dur, 90) * TICK * 1e6), "max": float(dur.max() * TICK * 1e6)},
        "nn_launch_us_overlapped_by_iteration": [round(float(x) * TICK * 1e6, 2) for x in (nn[:, :, 1] - nn[:, :, 0]).mean(axis=0)],
        "mean_resident_nn_kernels": conc_window,
        "mean_resident_nn_kernels_while_any": dur.sum() / busy if busy > 0 else None,
        "time_frac_with_n_resident": {str(k): round(v / window, 4) for k, v in sorted(level_t.items())},
        "sum_nn_us_per_alignment": sum_nn_per_align_us,
        "per_alignment_wall_us_device_clock": per_align_wall_us,
        "per_alignment_wall_us_host_clock": host_s * 1e6 / n_align,
        "identity": "sum_nn_us_per_alignment / mean_resident_nn_kernels == per_alignment_wall_us_device_clock (by construction); the "
                    "measured quantities are the two on the left, the host clock on the right is the independent check",
        "alignment_latency_us": {"mean": float(((a_end - a_start) * TICK * 1e6).mean()), "p90": float(np.percentile((a_end - a_start) * TICK * 1e6, 90))},
        "iterations_per_s_device_clock": P * iterations / (per_align_wall_us * 1e-6),
    }
    if ok_sv.any():
        sd = (sv[..., 1] - sv[..., 0])[ok_sv].astype(np.float64)
        out["solve_launch_us"] = {"mean": float(sd.mean() * TICK * 1e6), "launches_per_alignment": float(ok_sv.sum() / n_al)}
        # the gap between an iteration's NN end and the next iteration's NN start (solve launch + launch boundaries)
        gap = (nn[:, 1:, 0] - nn[:, :-1, 1]).astype(np.float64)
        out["nn_to_nn_gap_us"] = {"mean": float(gap.mean() * TICK * 1e6), "p90": float(np.percentile(gap, 90) * TICK * 1e6)}
    else:
        gap = (nn[:, 1:, 0] - nn[:, :-1, 1]).astype(np.float64)
        out["nn_to_nn_gap_us"] = {"mean": float(gap.mean() * TICK * 1e6), "p90": float(np.percentile(gap, 90) * TICK * 1e6)}
    return out


def h2d_rate(torch, dist, world, pool, dev, reps=24):
    """GB/s of pinned-host -> device copies of this rank's depth-image pool (all ranks copy at the same time)"""
    dst = torch.empty_like(pool.depth, device=dev)
    dst.copy_(pool.depth, non_blocking=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dst.copy_(pool.depth, non_blocking=True)
    torch.cuda.synchronize()
    return reps * pool.depth.numel() * 2 / (time.perf_counter() - t0) / 1e9


def timed_stream(torch, dist, world, streamer, steps, warmup, aligns_per_step, comm=None, P=1):
    """W untimed steps, then EXACTLY K steps between barrier + synchronize; the pose records of every step are
    gathered over RCCL (pipelined by one step, last table collected before the clock stops)."""
    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pending = 0
    table = None
    step_seeds = []          # seeds of this rank's records of the last step, in record order
    tg = [0.0, 0]            # host seconds inside the pose gather (submit + collect), calls

    class _Sink(list):
        pass

    def one_step():
        nonlocal pending, table, step_seeds
        sink = None
        if comm is not None:
            sink = _Sink(); sink.seeds = []
        res = streamer.run(aligns_per_step, sink)
        if sink is not None:
            step_seeds = sink.seeds
        if comm is not None:
            t0 = time.perf_counter()
            if pending >= 2:
                table = comm.gather_collect(aligns_per_step * P); pending -= 1
            comm.gather_submit(sink); pending += 1
            tg[0] += time.perf_counter() - t0; tg[1] += 1
        return res

    def drain():
        nonlocal pending, table
        t0 = time.perf_counter()
        while pending:
            table = comm.gather_collect(aligns_per_step * P); pending -= 1
        tg[0] += time.perf_counter() - t0

    res = None
    for _ in range(warmup):
        res = one_step()
    drain()
    fence()
    streamer.t_enqueue = streamer.t_fetch = 0.0; streamer.n_enqueued = 0
    tg[0], tg[1] = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(steps):
        res = one_step()
    drain()                 # the last step's pose table is on every rank before the clock stops
    fence()
    el = time.perf_counter() - t0
    diag = {"enqueue_us_per_alignment": 1e6 * streamer.t_enqueue / max(streamer.n_enqueued, 1),
            "fetch_wait_us_per_alignment": 1e6 * streamer.t_fetch / max(streamer.n_enqueued, 1),
            "gather_us_per_step": (1e6 * tg[0] / max(tg[1], 1)) if comm is not None else None, "timed_region_s": el}
    return el, res, table, step_seeds, diag


# ------------------------------------------------------------------------------------------------ rows f-1 / f-2
def seg_mode(args, torch, dist, capi, synth, world, rank, local_rank, dev):
    """Row f-2: plane segmentation of F = --pairs resident frames per GPU and step."""
    F = args.pairs
    seeds = [args.seed0 + rank * F + i for i in range(F)]
    frames = [synth.make_pair(s, args.width, args.height) for s in seeds]
    intr = frames[0].intr
    host = [synth.backproject_numpy(p.depth_src, intr).reshape(-1, 4) for p in frames]
    d = torch.from_numpy(np.stack(host)).to(dev)
    N = args.width * args.height
    d_lab = torch.zeros((F, N), dtype=torch.int32, device=dev)
    h = capi.IcpHandle(capi.default_params(intr, max_batch=F, device=local_rank))
    sp = h.seg_params(seed=args.seed0)
    ptrs = [d.data_ptr() + i * N * 16 for i in range(F)]
    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out_planes = None
    for _ in range(args.warmup):
        h.segment_planes_device(ptrs, sp, d_lab.data_ptr(), stream)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_planes = h.segment_planes_device(ptrs, sp, d_lab.data_ptr(), stream)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    rounds = [len(p) for p in out_planes]
    # point passes: init + per round (count, moments, label); each reads 16 B cloud + 4 B label per pixel
    passes = sum(1 + 3 * min(sp.max_planes, r + 1) for r in rounds)
    alg_bytes = passes * 20 * N
    value = world * F * args.steps / elapsed
    out = {
        "metric": f"plane segmentations/sec on {args.width}x{args.height} organized clouds", "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+int64", "data": "synthetic",
        "config": {"workload": f"row f-2: {F} frame(s) per GPU, <= {sp.max_planes} planes, {sp.hypotheses} hypotheses per round, "
                               f"threshold {sp.distance_threshold:.2f} m, plane_percent {sp.plane_percent:.1f}",
                   "frames_per_gpu": F, "planes_found": rounds[:8]},
        "roofline": {"kernel": "whole launch sequence (k_seg_init + rounds x {hyp+count, moments, refine+label})",
                     "bound": "hbm", "achieved": alg_bytes / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_step": alg_bytes,
                     "note": "a frame alone: 11 dependent small launches per call (init + 3 per round + final), launch latency bounds it; batches: 5 per round"},
    }
    if rank == 0:
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            t1 = time.perf_counter()
            po, lo = O.segment_planes(host[0], seed=args.seed0)
            dt = time.perf_counter() - t1
            lab0 = d_lab[0].cpu().numpy()
            out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "oracle/seg_oracle.c, frame 0, single thread"}
            out["parity_vs_oracle"] = {"labels_equal": bool(np.array_equal(lab0, lo)),
                                       "coeff_equal": bool(all(np.array_equal(a["coeff"], b["coeff"]) for a, b in zip(out_planes[0], po)))}
        _finish(out)
    h.close()


def voxel_mode(args, torch, dist, capi, synth, world, rank, local_rank, dev):
    """Row f-1: PassThrough + VoxelGrid(grid_leaf 0.03) of one resident 16-byte-record cloud per step."""
    pr = synth.make_pair(args.seed0 + rank, args.width, args.height)
    c = synth.backproject_numpy(pr.depth_src, pr.intr).reshape(-1, 4).copy()
    c[:, 3] = np.random.default_rng(args.seed0 + rank).integers(0, 2 ** 32, c.shape[0], dtype=np.uint64).astype(np.uint32).view(np.float32)
    n = c.shape[0]
    d = torch.from_numpy(c).to(dev)
    out_d = torch.zeros_like(d)
    h = capi.IcpHandle(capi.default_params(pr.intr, max_batch=1, device=local_rank))
    # a stream of the caller's: the call returns once the voxel count is known, the records follow in stream order, so the
    # next call's launches are queued while this one's tail still runs (the fence below drains the stream inside the timed region)
    ts = torch.cuda.Stream(device=dev)
    ts.wait_stream(torch.cuda.current_stream())
    stream = ts.cuda_stream

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    m = 0
    for _ in range(args.warmup):
        m = h.voxel_grid_device(d.data_ptr(), n, out_d.data_ptr(), 0.03, stream)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m = h.voxel_grid_device(d.data_ptr(), n, out_d.data_ptr(), 0.03, stream)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    alg_bytes = 16 * n + 16 * m
    per = elapsed / args.steps
    out = {
        "metric": f"voxel-grid down-samplings/sec of {args.width}x{args.height} clouds", "value": world * args.steps / elapsed,
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * per,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64 fixed point", "data": "synthetic",
        "config": {"workload": f"row f-1: PassThrough z<=7 + VoxelGrid leaf 0.03 on {n} records of 16 B -> {m} voxels", "points": n, "voxels": m},
        "roofline": {"kernel": "whole launch sequence of slam3d_voxel_grid_device", "bound": "hbm",
                     "achieved": alg_bytes / per / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_bytes / per / 1e9 / HBM_PEAK_GBPS,
                     "traffic": None, "algorithmic_bytes_per_step": alg_bytes,
                     "note": "algorithmic bytes = records in + records out"},
    }
    if rank == 0:
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            t1 = time.perf_counter()
            want = O.voxel_grid(c, 0.03, 7.0)
            dt = time.perf_counter() - t1
            got = out_d[:m].cpu().numpy()
            out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "oracle/voxel_oracle.c (qsort by voxel key), same cloud, single thread"}
            out["parity_vs_oracle"] = {"bit_identical": bool(got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)))}
        _finish(out)
    h.close()


# ------------------------------------------------------------------------------------------------ config 5
def dense_leg(args, torch, dist, capi, synth, world, rank, local_rank, comm, width, height, steps, warmup, want_cpu):
    """BASELINE config 5: ONE pair (seed 2000 + k), source rows sharded over the ranks, slam3d_icp_dense_run.  A step =
    one alignment including the H2D of both depth images (every rank uploads the whole pair)."""
    est = Est(capi, args.estimator, args.coarse_iterations)
    pool = Pool(torch, synth, [2000 + k for k in range(min(args.pool, 4))], width, height, args.noise_sigma, (args.hole_block, args.hole_prob))
    params = capi.default_params(pool.intr, iterations=args.iterations, max_batch=1, device=local_rank, **est.kw())
    h = capi.IcpHandle(params)
    k = [0]

    def step():
        j = k[0] % len(pool); k[0] += 1
        h.frame_set_depth_host_ptr(0, pool.src_ptr(j))
        h.frame_set_depth_host_ptr(1, pool.tgt_ptr(j))
        h.set_pair(0, 0, 1)
        return h.dense_run(comm)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(warmup):
        res = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    out = dict(elapsed=elapsed, res=res, pool=pool, handle=h, last_j=(k[0] - 1) % len(pool))
    # the roofline's launch time: the same loop again with per-launch HIP events (slam3d_icp_dense_run brackets every NN
    # launch when profiling is on); algorithmic bytes of THIS rank's share (its source rows, the whole target)
    h.set_profiling(True)
    nn, alg, flops = [], [], []
    pre = []
    for _ in range(min(6, max(2, steps))):
        r = step()
        nn.append(float(np.sum(h.get_iteration_timings())))
        pre.append(float(h.get_timings()["preprocess_ms"]))
        alg.append(alg_bytes_per_launch(r["n_src"], r["n_tgt"], est, args.iterations))
        flops.append(8.0 * r["n_src"] * r["n_tgt"])
    h.set_profiling(False)
    out["prof"] = dict(nn_ms=statistics.mean(nn), alg_bytes=statistics.mean(alg), flops=statistics.mean(flops))
    # SURVEY.md 8(e) / VERDICT r4 item 7c: the target's preprocessing is REPLICATED on every rank.  What it costs (measured here, HIP
    # events) against what sharding it would cost: the products a rank would have to receive from its peers -- normals 16 B, tile records
    # 18 B, image-order records 16 B, boxes: ~51 B per pixel -- over xGMI (ring all-gather, (N-1)/N of the bytes per rank at ~150 GB/s per
    # link pair) plus two collective latencies
    n_px = width * height
    prod_bytes = 51.0 * n_px
    out["target_preprocessing"] = {
        "replicated_ms_measured": statistics.mean(pre), "products_bytes": prod_bytes,
        "predicted_sharded_ms": {str(n): round(statistics.mean(pre) / n + prod_bytes * (n - 1) / n / 150e9 * 1e3 + 0.03, 4) for n in (2, 4, 8)},
        "note": "preprocessing (back-projection, normals, tiles of BOTH frames) stays replicated: recomputing costs less than gathering its products"}
    if world == 1:
        if want_cpu:
            j = out["last_j"]
            pr = pool.pairs[j]
            s4 = synth.backproject_numpy(pr.depth_src, pr.intr); t4 = synth.backproject_numpy(pr.depth_tgt, pr.intr)
            h.frame_set_depth_host_ptr(0, pool.src_ptr(j)); h.frame_set_depth_host_ptr(1, pool.tgt_ptr(j)); h.set_pair(0, 0, 1)
            r = h.dense_run(comm)
            idx, _ = h.get_correspondences(0)
            out["cpu"], out["parity"] = cpu_baseline_leg(pr, s4, t4, args.iterations, est, r, idx, brute_sample=False)
    return out


def main():
    args = parse_args()
    if args.legs_file:
        _LEGS_FILE[0] = os.path.abspath(args.legs_file)
    # the CPU baseline's OpenMP team must not keep spinning on the host's cores after its leg (libgomp's idle threads busy-wait
    # by default, and the legs that follow are driven from this process's Python threads): read when libgomp is first loaded
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import torch
    import torch.distributed as dist
    from slam3d_gx_amd import capi, synth

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched as a plain process: start the N ranks ourselves through the same launcher the driver uses
        # (one rank per GPU, rendezvous on 127.0.0.1) and hand its output and exit code through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the ICP path)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    host_comm = args.dist_backend == "gloo"       # tests only: several ranks on ONE GPU (RCCL needs one GPU per rank)
    comm = None
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if host_comm:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            # rendezvous of the library's own RCCL communicator: 128 bytes from rank 0 (torch.distributed is the host
            # side of the launch contract -- barrier, max-over-ranks -- the data-path exchanges run behind the C-ABI)
            uid = [capi.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            comm = _checked_comm(capi, torch, dist, uid[0], rank, world, local_rank, dev)

    if args.mode in ("seg", "voxel"):
        (seg_mode if args.mode == "seg" else voxel_mode)(args, torch, dist, capi, synth, world, rank, local_rank, dev)
        if comm is not None:
            comm.close()
        if dist.is_initialized():
            dist.destroy_process_group()
        _emit_final()
        return

    est = Est(capi, args.estimator, args.coarse_iterations)
    size_tag = f"{args.width}x{args.height}"
    metric = baseline_metric() if size_tag == "640x480" else f"ICP iterations/sec on {size_tag} clouds; SE(3) pose error vs PCL ref"

    def tmax(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device="cpu" if host_comm else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # ================================================================== --mode dense (config 5 as the main line)
    if args.mode == "dense":
        if host_comm and world > 1:
            raise SystemExit("--mode dense needs RCCL (one GPU per rank); the gloo form is covered by tests/test_shard_gloo.py")
        if world > 1 and comm is None:
            # without the library's communicator slam3d_icp_dense_run would align the WHOLE pair on every rank while the
            # line below says "sharded": refuse instead of mislabelling
            raise SystemExit("--mode dense: the C-side RCCL communicator is unavailable (slam3d_comm self-test failed); "
                             "refusing to run unsharded under a 'sharded' label")
        d = dense_leg(args, torch, dist, capi, synth, world, rank, local_rank, comm, args.width, args.height, args.steps, args.warmup,
                      want_cpu=(rank == 0 and world == 1 and not args.no_cpu_baseline))
        elapsed = tmax(d["elapsed"])
        out = {
            "metric": metric, "value": args.iterations * args.steps / elapsed, "unit": "ICP iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 5: one {size_tag} pair per step (seeds 2000..), source rows sharded over {world} GPU(s), "
                                   f"{args.iterations} iterations, {args.estimator}; H2D of both u16 depth images inside every step; "
                                   "one in-place ncclAllReduce of the iteration's integer Gram totals (16 replicas x 40 int64) inside slam3d_icp_dense_run",
                       "iterations": args.iterations, "estimator": args.estimator,
                       "parallelism": f"source rows over {world} rank(s), RCCL all-reduce (C-ABI) of 16 x 40 int64 per iteration",
                       "coarse_iterations": args.coarse_iterations,
                       "n_src": d["res"]["n_src"], "n_tgt": d["res"]["n_tgt"]},
            "status": [d["res"]["status"]],
            # what this mode is DESIGNED to reach: the target's preprocessing and the H2D are replicated on
            # every rank (~0.25 ms of a 1.97 ms alignment), a launch has a ~20 us latency floor however few rows it holds, and
            # every iteration adds one 4 KB all-reduce (~15-25 us): a latency experiment, not the scaling mode -- read a
            # measured ratio against THIS, not against the >= 6x target of the batch mode (no data-path collective)
            "predicted_scaling": {"vs_1_gpu": {"2": 1.5, "4": 2.0, "8": 2.2},
                                  "model": "T(N) = 0.25 ms (H2D + preprocessing, replicated) + iterations x (max(88 us / N, 20 us) + "
                                           "allreduce 15..25 us + head solve 2..3 us); T(1) = 1.97 ms measured"},
        }
        if "prof" in d:
            out["roofline"] = tiles_roofline(d["prof"], args.iterations, f"k_nn_tiles_acc_{size_tag}")
            out["target_preprocessing"] = d.get("target_preprocessing")
        if "cpu" in d:
            out["cpu_baseline"], out["parity_vs_oracle"] = d["cpu"], d["parity"]
        if rank == 0:
            _finish(out)
        d["handle"].close()
        if comm is not None:
            comm.close()
        if dist.is_initialized():
            dist.destroy_process_group()
        _emit_final()
        return

    # ================================================================== batch mode: configs 2 / 3
    P = args.pairs
    S = max(P, (args.pairs_per_step // P) * P)
    aligns = S // P
    n_handles = 1 if args.no_pipeline else max(1, min(16, args.in_flight))
    pool_n = max(args.pool, P)
    # every in-flight handle streams its own distinct pairs: seeds seed0 + (rank * n_handles + hi) * pool_n + k
    want_extra = (rank == 0 and world == 1 and not args.no_extra_configs and args.nn_mode in (capi.NN_AUTO, capi.NN_TILES)
                  and size_tag == "640x480" and P == 1)
    mask = (args.hole_block, args.hole_prob)
    specs = [Pool.spec(args.seed0 + (rank * n_handles + hi) * pool_n + k, args.width, args.height, args.noise_sigma, mask)
             for hi in range(n_handles) for k in range(pool_n)]
    headline_is_baseline_md = abs(args.noise_sigma - 0.0012) < 1e-9 and mask == (8, 0.25)
    if want_extra:       # render everything the extra legs need in the same parallel batch
        specs += [Pool.spec(args.seed0 + k, 640, 480, args.noise_sigma, mask) for k in range(64)]
        leg8 = [(hi, k) for hi in range(n_handles) for k in range(min(pool_n, 8))]
        if headline_is_baseline_md:
            specs += [Pool.spec(args.seed0 + hi * pool_n + k, 640, 480, 0.0002, (32, 0.2)) for hi, k in leg8]
        else:
            specs += [Pool.spec(args.seed0 + hi * pool_n + k, 640, 480, 0.0012, mask) for hi, k in leg8]
            specs += [Pool.spec(args.seed0 + hi * pool_n + k, 640, 480, args.noise_sigma, (8, 0.25)) for hi, k in leg8]
            specs += [Pool.spec(args.seed0 + hi * pool_n + k, 640, 480, 0.0012, (8, 0.25)) for hi, k in leg8]
        specs += [Pool.spec(2000 + k, 1280, 960, args.noise_sigma, mask) for k in range(min(args.pool, 4))]
    Pool.prefetch(synth, specs)
    pools = [Pool(torch, synth, [args.seed0 + (rank * n_handles + hi) * pool_n + k for k in range(pool_n)], args.width, args.height,
                  args.noise_sigma, mask) for hi in range(n_handles)]
    intr = pools[0].intr
    params = capi.default_params(intr, iterations=args.iterations, max_batch=P, device=local_rank, nn_mode=args.nn_mode, **est.kw())
    handles = [capi.IcpHandle(params) for _ in range(n_handles)]
    streamer = Streamer(handles, pools, P)
    gather = None
    if comm is not None:
        gather = comm
    elif world > 1 and (host_comm or _COMM_FALLBACK[0]):
        from slam3d_gx_amd import shard

        class _HostGather:        # gloo (tests), or the RCCL fallback of _checked_comm: same submit/collect contract through torch.distributed
            def __init__(self):
                self.g = shard.PoseGatherer(world * S, device=None if host_comm else dev)

            def gather_submit(self, results):
                self.g.submit(shard.pack_records(results))

            def gather_collect(self, n):
                return shard.unpack_records(self.g.collect())
        gather = _HostGather()
    pose_exchange = ("none (1 rank)" if gather is None else "rccl: ncclAllGather behind the C-ABI (slam3d_pose_gather_*)" if comm is not None
                     else "gloo through torch.distributed (tests)" if host_comm else "torch.distributed fallback (slam3d_comm self-test failed)")
    elapsed, res, table, step_seeds, diag = timed_stream(torch, dist, world, streamer, args.steps, args.warmup, aligns, gather, P)
    elapsed = tmax(elapsed)
    # per-rank attribution of a sub-linear scaling curve: this rank's host-to-device rate with nothing else
    # running (all ranks measure at once: they share the host's PCIe root / memory controllers exactly as in the timed region),
    # host time to queue one alignment, host time blocked in the fetch, host time inside the pose gather
    diag["rank"] = rank
    diag["h2d_GBps"] = h2d_rate(torch, dist, world, pools[0], dev)
    per_rank = [diag]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, diag)
    total_iters = world * S * args.iterations * args.steps
    value = total_iters / elapsed

    out = {
        "metric": metric, "value": value, "unit": "ICP iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (f"BASELINE config {'2' if P == 1 else '3'}: a step = {S} frame pairs of {size_tag} per GPU as {aligns} consecutive "
                         f"alignment(s) of {P} pair(s), {args.iterations} ICP iterations each, {args.estimator}, exact NN (tile-pruned brute "
                         f"force); EVERY alignment uploads both u16 depth images of its pair(s) from pinned host memory and rebuilds "
                         f"normals + tiles (nothing cached between alignments); pairs cycle through {pool_n} distinct seeds per in-flight "
                         f"handle (seeds {pools[0].pairs[0].seed}..{pools[-1].pairs[-1].seed}), noise sigma {args.noise_sigma} z^2, invalid-pixel blocks "
                         f"{args.hole_block}x{args.hole_block} px with p = {args.hole_prob}"),
            "pairs_per_step_per_gpu": S, "pairs_per_launch": P, "alignments_per_step": aligns, "iterations": args.iterations,
            "estimator": args.estimator, "distinct_pairs_per_handle": pool_n, "noise_sigma_over_z2": args.noise_sigma,
            "hole_block_px": args.hole_block, "hole_prob": args.hole_prob,
            "synthetic_workload": ("BASELINE.md section 4 as specified" if headline_is_baseline_md else "NOT BASELINE.md section 4's (sigma 0.0012 z^2, 8x8 holes at p 0.25)"),
            "coarse_iterations": max(0, min(args.coarse_iterations, args.iterations - 1)),
            "coarse_iterations_note": ("spec S4c: that many leading iterations of every run take the sources of every fourth 8x8-pixel tile only; they count "
                                       "as iterations in `value` (rounds 1-3 ran every iteration on every source: compare rounds with --coarse-iterations 0, "
                                       "leg `all_sources_every_iteration`)"),
            "h2d_bytes_per_pair": 2 * pools[0].frame_bytes,
            "step_pipelining": (f"{n_handles} handles, each on its own HIP stream, take turns: alignment k+1.. are queued (H2D + kernels) before "
                                f"alignment k's poses are fetched; every pose reaches the host inside the timed region") if n_handles > 1 else "none",
            "nn_mode": {0: "auto(tiles)", 1: "brute_valu", 2: "brute_mfma", 3: "tiles"}.get(args.nn_mode, str(args.nn_mode)), "in_flight": n_handles,
            "n_src": [r["n_src"] for r in res][:4], "n_tgt": [r["n_tgt"] for r in res][:4],
            "parallelism": (f"pairs sharded one process per GPU x{world}; one RCCL all-gather (C-ABI slam3d_pose_gather_*) of the step's "
                            f"{S} pose records per rank, overlapping the next step" if world > 1 else "1 GPU"),
        },
        "status": [r["status"] for r in res][:8],
        "timed_region_s": elapsed,
        "rccl_ranks": (comm.world if comm is not None else 0), "pose_exchange": pose_exchange,
        "per_rank": per_rank,
    }
    if table is not None:
        out["config"]["gathered_pose_records"] = len(table)
        # which pair each gathered record belongs to (outside the timed region): rank blocks in rank order = pair order
        all_seeds = [step_seeds]
        if world > 1:
            all_seeds = [None] * world
            dist.all_gather_object(all_seeds, step_seeds)
        flat = [sd for blk in all_seeds for sd in blk]
        out["config"]["gathered_seeds"] = {"first": flat[:2], "last": flat[-2:], "ascending": bool(all(b > a for a, b in zip(flat, flat[1:])))}
        if args.dump_table and rank == 0:
            with open(args.dump_table, "w") as f:
                json.dump({"seeds": flat, "world": world, "pairs_per_rank": len(step_seeds),
                           "T": [np.asarray(r["T"], dtype=np.float64).reshape(16).tolist() for r in table],
                           "inliers": [int(r["inliers"]) for r in table], "status": [int(r["status"]) for r in table],
                           "norm": [float(r["norm"]) for r in table]}, f)

    tiles = args.nn_mode in (capi.NN_AUTO, capi.NN_TILES)
    if args.timed_only:
        if rank == 0:
            _finish(out)
        for hh in handles:
            hh.close()
        if comm is not None:
            comm.close()
        if dist.is_initialized():
            dist.destroy_process_group()
        _emit_final()
        return
    # ---- un-overlapped latency and the event-profiled pass (rank-local; reported by rank 0)
    st1 = Streamer([handles[0]], [pools[0]], P)
    nl = min(32, max(4, aligns))
    st1.run(2)
    torch.cuda.synchronize()
    tl = time.perf_counter()
    st1.run(nl)
    torch.cuda.synchronize()
    out["single_step_latency_ms"] = 1e3 * (time.perf_counter() - tl) / nl
    out["single_step_latency_note"] = f"one alignment ({P} pair(s)) at a time incl. H2D of both depth images, mean of {nl}"
    if tiles and n_handles > 1 and args.overlap_aligns > 0:
        out["overlap"] = overlap_pass(torch, handles, pools, P, max(8 * n_handles, min(args.overlap_aligns, 4 * aligns)), args.iterations)
        out["overlap"]["value_unstamped"] = value / world
    prof = profiled_pass(handles[0], pools[0], P, max(4, min(args.profile_aligns, 4 * aligns)), est)
    out["kernel_ms_per_alignment"] = {"preprocess": prof["preprocess_ms"], "nn": prof["nn_ms"], "total": prof["total_ms"]}
    out["kernel_only_value"] = P * args.iterations / (prof["total_ms"] * 1e-3)
    out["nn_ms_per_iteration"] = prof["per_it"]
    if tiles:
        out["roofline"] = tiles_roofline(prof, args.iterations, f"k_nn_tiles_acc_{size_tag}_P{P}")
    else:
        launch_ms = prof["nn_ms"] / max(args.iterations, 1)
        ach = prof["flops"] / (launch_ms * 1e-3) / 1e12
        out["roofline"] = {"kernel": "k_nn_mfma" if args.nn_mode == capi.NN_BRUTE_MFMA else "k_nn_valu (full brute-force scan)",
                           "bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS,
                           "traffic": None, "launch_ms": launch_ms, "flops_per_launch": prof["flops"]}

    if rank == 0 and world == 1:
        pr0 = pools[0].pairs[0]
        s4 = synth.backproject_numpy(pr0.depth_src, intr); t4 = synth.backproject_numpy(pr0.depth_tgt, intr)
        if tiles and not args.no_bruteforce:
            out["roofline_bruteforce"] = bruteforce_leg(capi, intr, est, s4, t4, local_rank)
        if not args.no_cpu_baseline:
            h0 = handles[0]
            h0.frame_set_depth_host_ptr(0, pools[0].src_ptr(0)); h0.frame_set_depth_host_ptr(1, pools[0].tgt_ptr(0)); h0.set_pair(0, 0, 1)
            h0.run(1)
            r0 = h0.fetch_results(1)[0]
            idx, _ = h0.get_correspondences(0)
            out["cpu_baseline"], out["parity_vs_oracle"] = cpu_baseline_leg(pr0, s4, t4, args.iterations, est, r0, idx)
        if want_extra:
            extra_legs(args, torch, dist, capi, synth, local_rank, est, handles, pools, out)
    if rank == 0:
        # the flat report SURVEY.md 8(d) lists, assembled from the objects above (for N > 1 the single-GPU legs -- CPU
        # baseline, brute-force rooflines, parity -- are not re-run and read null)
        rb, rf, cb_, pv = out.get("roofline_bruteforce", {}), out.get("roofline", {}), out.get("cpu_baseline", {}), out.get("parity_vs_oracle") or {}
        out["survey_8d"] = {
            "gpus": world, "pairs": world * S * args.steps, "iters": args.iterations, "wall_s": elapsed,
            "icp_iters_per_s": value, "kernel_only_icp_iters_per_s": out["kernel_only_value"],
            "nn_tflops": rb.get("equivalent_f32_contraction_tflops"), "nn_frac_fp32_peak": rb.get("equivalent_frac_of_f32_peak"),
            "nn_bf16_tflops_executed": rb.get("achieved"), "nn_frac_bf16_peak": rb.get("frac"),
            "hbm_GBps": rf.get("achieved") if rf.get("unit") == "GB/s" else None,
            "hbm_frac_peak": rf.get("frac") if rf.get("unit") == "GB/s" else None,
            "cpu_A_iters_per_s": cb_.get("bruteforce_value"), "cpu_B_iters_per_s": cb_.get("value"),
            "cpu_B_1thread_iters_per_s": cb_.get("single_thread_value"), "cores": cb_.get("cores"),
            "max_rot_err": pv.get("rot_err_rad"), "max_trans_err": pv.get("trans_err_m"), "idx_mismatches": pv.get("idx_mismatches"),
        }
    if rank == 0:
        _finish(out)
    for hh in handles:
        hh.close()
    if comm is not None:
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    _emit_final()


_COMM_FALLBACK = [False]


def _checked_comm(capi, torch, dist, uid, rank, world, local_rank, dev, timeout_s=90.0):
    """The library's own RCCL communicator, created and exercised (one tiny pose gather) in a helper thread with a deadline.
    This box has one GPU, so the multi-rank form of slam3d_comm_* first runs on the driver's node: if it does not answer in
    time on ANY rank, every rank falls back to the same exchange through torch.distributed (slam3d_gx_amd/shard.py) instead
    of hanging the run, and the JSON line says so (`config.pose_exchange`)."""
    import threading
    box = {}

    def work():
        try:
            c = capi.Comm(uid, rank, world, local_rank)
            r = c.gather([dict(T=np.eye(4) * (rank + 1), norm=float(rank), inliers=rank, status=0, rmse=0.0)])
            if len(r) == world and all(int(r[k]["inliers"]) == k for k in range(world)):
                box["comm"] = c
        except Exception as e:      # noqa: BLE001 -- reported below
            box["error"] = repr(e)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout_s)
    ok = torch.tensor([1 if ("comm" in box and not t.is_alive()) else 0], dtype=torch.int32, device=dev)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        return box["comm"]
    _COMM_FALLBACK[0] = True
    if rank == 0:
        print(f"bench.py: slam3d_comm self-test failed ({box.get('error', 'no answer within %.0f s' % timeout_s)}); "
              "pose records go through torch.distributed instead", file=sys.stderr)
    return None


def voxel_icp_leg(args, torch, capi, synth, local_rank, vh, d1, d2, want_cpu):
    """Frame ingestion + alignment as the reference runs them (src/GraphicEnd.cpp:279-295 then :158,:168): the two frames' 16-byte
    records resident on the device -> PassThrough z <= 7 + VoxelGrid(0.03) on the device (vh: a 640x480 handle's tables) ->
    20 ICP iterations between the two voxel clouds on an unorganized handle -> pose on the host.  One alignment at a time."""
    dev = f"cuda:{local_rank}"
    kintr = synth.Intrinsics()
    recs = []
    for d in (d1, d2):
        c = synth.backproject_numpy(d, kintr, z_filter=1e9).reshape(-1, 4)
        c = c[np.isfinite(c[:, 2])].copy()
        c[:, 3] = np.float32(0)
        recs.append(c)
    d_in = [torch.from_numpy(c).to(dev) for c in recs]
    W = 16384
    d_vox = [torch.full((W, 4), float("nan"), dtype=torch.float32, device=dev) for _ in recs]
    stream = torch.cuda.current_stream().cuda_stream
    uintr = synth.Intrinsics(width=W, height=1)
    Ti = synth.pose_from_seed(77, 2.0, 0.03)
    res = {}
    with capi.IcpHandle(capi.default_params(uintr, iterations=args.iterations, estimator=capi.EST_SVD, device=local_rank)) as h:
        def align(a, b, T0, voxelise=True):
            ms = [0, 0]
            if voxelise:
                ms[0] = vh.voxel_grid_device(d_in[a].data_ptr(), len(recs[a]), d_vox[0].data_ptr(), 0.03, stream)
                ms[1] = vh.voxel_grid_device(d_in[b].data_ptr(), len(recs[b]), d_vox[1].data_ptr(), 0.03, stream)
                torch.cuda.synchronize()        # (the ICP handle runs on its own stream: the records must be complete)
            h.set_clouds_device(0, d_vox[0].data_ptr(), d_vox[1].data_ptr())
            h.run(1, None if T0 is None else T0.reshape(1, 16))
            return h.fetch_results(1)[0], ms
        for name, a, b, T0 in (("dep1_to_dep2", 0, 1, None), ("dep1_to_dep1_perturbed", 0, 0, Ti)):
            d_vox[0].fill_(float("nan")); d_vox[1].fill_(float("nan"))
            r, ms = align(a, b, T0)
            for _ in range(3):
                align(a, b, T0)
            torch.cuda.synchronize()
            n = 40
            t0 = time.perf_counter()
            for _ in range(n):
                r, ms = align(a, b, T0)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                r, _ = align(a, b, T0, voxelise=False)
            torch.cuda.synchronize()
            dt_icp = (time.perf_counter() - t0) / n
            leg = {"points": ms, "value": args.iterations / dt, "unit": "ICP iterations/s incl. the voxel grid of both frames", "ms_per_alignment": 1e3 * dt,
                   "icp_only_value": args.iterations / dt_icp, "icp_only_ms": 1e3 * dt_icp, "us_per_iteration": 1e6 * dt_icp / args.iterations, "inliers": r["inliers"], "status": r["status"], "norm": r["norm"]}
            if want_cpu:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib as O
                va, vb = d_vox[0].cpu().numpy().reshape(1, W, 4), d_vox[1].cpu().numpy().reshape(1, W, 4)
                ro = O.icp(va, vb, O.params(uintr, iterations=args.iterations, estimator=1, nn_method=1, threads=min(16, os.cpu_count() or 1)), T_init=T0)
                # the headline's protocol: every thread count in its own process, team pinned to physical cores of one
                # NUMA node, median of 7 after a warm-up, fastest count among the steady ones
                curve, bc = cpu_thread_curve(va, vb, W, 1, dict(estimator=1), args.iterations, (1, 8, 16, 32, 64), 7, T_init=T0)
                leg["cpu_baseline"] = {"value": curve[bc]["value"], "unit": "ICP iterations/s", "cores": bc, "kind": "port",
                                       "median_value": curve[bc]["value"], "min_value": curve[bc]["min_value"], "spread": curve[bc]["spread"], "placement": curve[bc]["placement"],
                                       "thread_curve": {str(k): round(v["value"], 1) for k, v in curve.items()},
                                       "thread_curve_spread": {str(k): round(v["spread"], 3) for k, v in curve.items()},
                                       "sample": "oracle kd-tree ICP (svd) on the same two voxel clouds, threads pinned to physical cores (one NUMA node first; own "
                                                 "process), median of 7 after a warm-up; fastest of 1 / 8 / 16 / 32 / 64 threads with IQR spread <= 20 %"}
                leg["vs_cpu"] = leg["icp_only_value"] / curve[bc]["value"]
                leg["parity_vs_oracle"] = {"T_bit_identical": bool(np.array_equal(ro["T_trace"][-1], r["T_raw"])), "inliers_equal": bool(ro["inliers"] == r["inliers"])}
            res[name] = leg
    # the batched voxel grid (VERDICT r4 'Missing 4'): B copies of frame 1's records in ONE launch sequence
    B = 64
    outs = [torch.empty((len(recs[0]), 4), dtype=torch.float32, device=dev) for _ in range(B)]
    pin = [d_in[0].data_ptr()] * B
    pout = [o.data_ptr() for o in outs]
    nn_ = [len(recs[0])] * B
    ms = vh.voxel_grid_batch_device(pin, nn_, pout, 0.03, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        ms = vh.voxel_grid_batch_device(pin, nn_, pout, 0.03, stream)
    torch.cuda.synchronize()
    dtb = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(64):
        m1 = vh.voxel_grid_device(d_in[0].data_ptr(), len(recs[0]), outs[0].data_ptr(), 0.03, stream)
    torch.cuda.synchronize()
    dt1 = (time.perf_counter() - t0) / 64
    alg = B * 16.0 * (len(recs[0]) + ms[0])
    res["voxel_grid_batch"] = {"frames": B, "records_per_frame": len(recs[0]), "voxels": ms[0], "frames_per_s": B / dtb, "us_per_frame": 1e6 * dtb / B,
                               "single_call_us": 1e6 * dt1, "single_call_voxels": m1,
                               "roofline": {"bound": "hbm", "achieved": alg / dtb / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / dtb / 1e9 / HBM_PEAK_GBPS,
                                            "algorithmic_bytes_per_step": alg, "note": "records in + records out, 64 clouds per launch sequence"}}
    res["note"] = ("the reference's operating point: readimage's voxel clouds (16,034 / 14,758 points) aligned as unorganized point lists, svd estimator, "
                   "ONE persistent launch per run (csrc/list_icp.hpp: Morton-cell tiles, exact tile-pruned search, grid barrier); parity: tests/test_unorganized.py")
    return res


def extra_legs(args, torch, dist, capi, synth, local_rank, est, handles, pools, out):
    """Same process, same box, N = 1: the other regimes next to the headline, each measured (never derived)."""
    intr = pools[0].intr
    # ---- (i) the former headline: ONE pair resident in HBM, re-run in place (frames re-declared each time, so the
    # preprocessing still runs; no H2D, identical frame every time -> primed ownership map and caches)
    pr0 = pools[0].pairs[0]
    s4 = synth.backproject_numpy(pr0.depth_src, intr); t4 = synth.backproject_numpy(pr0.depth_tgt, intr)
    d_s = torch.from_numpy(s4).to(f"cuda:{local_rank}"); d_t = torch.from_numpy(t4).to(f"cuda:{local_rank}")
    n = 600

    def resident(k):
        q = []
        for i in range(k):
            h = handles[i % len(handles)]
            if len(q) >= len(handles):
                q.pop(0).fetch_results(1)
            h.set_clouds_device(0, d_s.data_ptr(), d_t.data_ptr())
            h.run(1)
            q.append(h)
        while q:
            q.pop(0).fetch_results(1)
    resident(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    resident(n)
    torch.cuda.synchronize()
    out["resident_same_pair_value"] = n * args.iterations / (time.perf_counter() - t0)
    out["resident_same_pair_note"] = (f"round-1 headline regime: the same pair (seed {pr0.seed}) resident in HBM, {n} runs in place, "
                                      f"{len(handles)} in flight, no H2D; best case, not the metric")
    # ---- (i-b) the front end's steady state (GraphicEnd::run, src/GraphicEnd.cpp:168): the keyframe stays the SOURCE of many
    # consecutive alignments, only the present frame is new.  Keyframes resident as frames of the handle (uploaded and
    # preprocessed once, before the clock starts); every alignment uploads ONE depth image and preprocesses one frame.
    def keyframe_leg(ke, n_align=1024):
        kf_params = capi.default_params(intr, iterations=args.iterations, max_batch=1, device=local_rank, extra_frames=len(pools[0]), **ke.kw())
        kf_handles = [capi.IcpHandle(kf_params) for _ in handles]
        try:
            for hh, pool in zip(kf_handles, pools):
                for j in range(len(pool)):
                    hh.frame_set_depth_host_ptr(hh.first_free_frame() + j, pool.src_ptr(j))

            def kf_stream(k):
                q = []
                for i in range(k):
                    hi = i % len(kf_handles)
                    hh, pool = kf_handles[hi], pools[hi]
                    if len(q) >= len(kf_handles):
                        q.pop(0).fetch_results(1)
                    j = (i // len(kf_handles)) % len(pool)
                    hh.frame_set_depth_host_ptr(1, pool.tgt_ptr(j))
                    hh.set_pair(0, hh.first_free_frame() + j, 1)
                    hh.run(1)
                    q.append(hh)
                while q:
                    q.pop(0).fetch_results(1)
            kf_stream(2 * len(pools[0]) * len(kf_handles))          # every keyframe has been a source once: its source-side products (and planes) exist
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kf_stream(n_align)
            torch.cuda.synchronize()
            return n_align * args.iterations / (time.perf_counter() - t0)
        finally:
            for hh in kf_handles:
                hh.close()
    out["keyframe_resident_value"] = keyframe_leg(est)
    out["keyframe_resident_note"] = ("the front end's steady state: keyframes (sources) resident as frames of the handle, uploaded and "
                                     "preprocessed once; every alignment uploads only the present frame's depth image (614 KB) and "
                                     f"preprocesses that one frame; 1024 alignments, {len(handles)} in flight, {len(pools[0])} keyframes per handle")
    # ---- (ii) survey noise level sigma = 0.0012 z^2, same streaming regime
    if abs(args.noise_sigma - 0.0012) > 1e-9:
        pools12 = [Pool(torch, synth, [p.seed for p in pool.pairs[:8]], args.width, args.height, 0.0012) for pool in pools]
        st = Streamer(handles, pools12, 1)
        st.run(32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 768
        st.run(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        v = k * args.iterations / dt
        out["survey_noise"] = {"noise_sigma_over_z2": 0.0012, "value": v, "ratio_to_headline": v / out["value"], "alignments": k,
                               "n_tgt": st.last[0]["n_tgt"], "inliers": st.last[0]["inliers"],
                               "note": "same streaming regime (H2D + distinct pairs); with iid noise at this level the reference's 0.01 m / "
                                       "41-of-49 planarity rule keeps far fewer target normals (DESIGN.md section 3, deviation ii)"}
    # ---- (ii-b) the invalid-pixel mask SURVEY.md 8(d) specifies: 8x8-pixel Bernoulli holes at p = 0.25 (four to five times the
    # hole-border length of the default 32x32 / 0.2 mask), same streaming regime; launch time from its own event-profiled pass
    def stream_leg(hs, lpools, k, T_init=None, warm=32):
        st = Streamer(hs, lpools, 1, T_init=T_init)
        st.run(warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st.run(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st1 = Streamer([hs[0]], [lpools[0]], 1, T_init=T_init)
        st1.run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st1.run(16)
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 16
        hs[0].set_profiling(True)
        nn, pre, per_it = [], [], None
        for _ in range(6):
            st1.run(1)
            tm = hs[0].get_timings()
            nn.append(tm["nn_ms"]); pre.append(tm["preprocess_ms"])
            per_it = hs[0].get_iteration_timings()
        hs[0].set_profiling(False)
        v = k * args.iterations / dt
        return st.last[0], {"value": v, "ratio_to_headline": v / out["value"], "alignments": k, "single_step_latency_ms": 1e3 * lat,
                            "nn_launch_us": 1e3 * statistics.mean(nn) / max(args.iterations, 1), "preprocess_us": 1e3 * statistics.mean(pre),
                            "nn_launch_us_by_iteration": [round(1e3 * float(x), 1) for x in per_it]}

    if (args.hole_block, args.hole_prob) != (8, 0.25):
        pools_m = [Pool(torch, synth, [p.seed for p in pool.pairs[:8]], args.width, args.height, args.noise_sigma, (8, 0.25)) for pool in pools]
        r, leg = stream_leg(handles, pools_m, 768)
        leg.update({"hole_block": 8, "hole_prob": 0.25, "n_src": r["n_src"], "n_tgt": r["n_tgt"], "inliers": r["inliers"], "status": r["status"],
                    "note": "SURVEY.md 8(d)'s mask (8x8-pixel Bernoulli holes, p = 0.25) in the same streaming regime (H2D + distinct pairs, "
                            f"{len(handles)} in flight); nn_launch_us by HIP events one alignment at a time"})
        out["survey_mask"] = leg
    # ---- (ii-b2) BASELINE.md section 4 / SURVEY.md 8(d)'s workload AS SPECIFIED, both parts together: sigma = 0.0012 z^2 AND the
    # 8x8 / 0.25 mask (only ever measured separately).  Parity of this workload: tests/test_gpu_parity.py::
    # test_full_640x480_baseline_md_workload (every iterate and index against the kd-tree oracle).
    if not (abs(args.noise_sigma - 0.0012) < 1e-9 and (args.hole_block, args.hole_prob) == (8, 0.25)):
        pools_b = [Pool(torch, synth, [p.seed for p in pool.pairs[:8]], args.width, args.height, 0.0012, (8, 0.25)) for pool in pools]
        r, leg = stream_leg(handles, pools_b, 768)
        leg.update({"noise_sigma_over_z2": 0.0012, "hole_block": 8, "hole_prob": 0.25, "n_src": r["n_src"], "n_tgt": r["n_tgt"],
                    "inliers": r["inliers"], "status": r["status"],
                    "note": "BASELINE.md section 4's synthetic workload exactly (sigma = 0.0012 z^2, 8x8-pixel Bernoulli holes at p = 0.25), same "
                            f"streaming regime (H2D + distinct pairs, {len(handles)} in flight); with iid noise at this level the reference's "
                            "0.01 m / 41-of-49 planarity rule keeps a normal on about a fifth of the targets (n_tgt)"})
        out["baseline_md_workload"] = leg
    # ---- (ii-b3) round 5: the headline IS BASELINE.md's workload; the low-noise surrogate that rounds 1-4 quoted is the leg
    if abs(args.noise_sigma - 0.0012) < 1e-9 and (args.hole_block, args.hole_prob) == (8, 0.25):
        pools_s = [Pool(torch, synth, [p.seed for p in pool.pairs[:8]], args.width, args.height, 0.0002, (32, 0.2)) for pool in pools]
        r, leg = stream_leg(handles, pools_s, 768)
        leg.update({"noise_sigma_over_z2": 0.0002, "hole_block": 32, "hole_prob": 0.2, "n_src": r["n_src"], "n_tgt": r["n_tgt"], "inliers": r["inliers"],
                    "status": r["status"], "note": "the low-noise surrogate rounds 1-4 quoted as the headline (an iid stand-in for correlated Kinect noise: "
                                                   "nearly every target keeps a 7x7-window normal), same streaming regime"})
        out["low_noise_surrogate"] = leg
    # ---- (ii-b4) every iteration on every source (coarse_iterations = 0: what rounds 1-3 ran and SURVEY.md App. C3 literally says)
    if est.coarse > 0:
        p0 = capi.default_params(intr, iterations=args.iterations, max_batch=1, device=local_rank, **dict(est.kw(), coarse_iterations=0))
        h0s = [capi.IcpHandle(p0) for _ in handles]
        try:
            r, leg = stream_leg(h0s, pools, 768)
            leg.update({"coarse_iterations": 0, "n_src": r["n_src"], "n_tgt": r["n_tgt"], "inliers": r["inliers"], "status": r["status"],
                        "note": "the headline stream with coarse_iterations = 0 (all 20 iterations on every source)"})
            out["all_sources_every_iteration"] = leg
        finally:
            for hh in h0s:
                hh.close()
    # ---- (ii-b5) round 5: SLAM3D_EST_PLANE -- plane-ICP proper (the frames' planes give the normals; + the plane-pair gate), same
    # streaming regime on the headline pairs; every alignment also segments both frames (target normals, source labels)
    if est.lib != capi.EST_PLANE:
        pe = Est(capi, "plane_gate", est.coarse)
        pp_ = capi.default_params(intr, iterations=args.iterations, max_batch=1, device=local_rank, **pe.kw())
        hps = [capi.IcpHandle(pp_) for _ in handles]
        try:
            r, leg = stream_leg(hps, pools, 512)
            leg.update({"estimator": "SLAM3D_EST_PLANE + SLAM3D_PLANE_PAIR_GATE", "n_src": r["n_src"], "n_tgt": r["n_tgt"], "inliers": r["inliers"],
                        "status": r["status"], "kernel_ms_per_alignment": leg.pop("kernel_ms", None),
                        "note": "spec S2p / S4p: both frames segmented on the device inside every alignment (3 RANSAC rounds of 64 hypotheses), pixels "
                                "on a plane take its least-squares normal, the others their 7x7-window normal; correspondences only inside "
                                "associated plane pairs; parity: tests/test_plane_icp.py"})
            out["plane_normals"] = leg
        finally:
            for hh in hps:
                hh.close()
        # the front end's steady state under the plane estimator: keyframes resident (segmented ONCE, their planes cached with the frame),
        # every alignment uploads, back-projects and segments only the present frame
        v = keyframe_leg(pe, 512)
        out["plane_normals"]["keyframe_resident_value"] = v
        out["plane_normals"]["keyframe_resident_ratio_to_headline"] = v / out["value"]
    # ---- (ii-c) real sensor frames: the reference's Kinect depth images (tests/golden/kinect, data fixtures).  dep1 -> dep2 is a
    # wide-baseline pair (an equality test elsewhere, here only a timing on real hole / edge geometry); dep_k -> dep_k from a small
    # initial guess converges to the identity
    kin = os.path.join(ROOT, "tests", "golden", "kinect")
    try:
        from PIL import Image
        d1 = np.array(Image.open(os.path.join(kin, "exp1_dep_1.png"))).astype(np.uint16)
        d2 = np.array(Image.open(os.path.join(kin, "exp1_dep_2.png"))).astype(np.uint16)
    except Exception as e:      # noqa: BLE001 -- the leg is optional
        d1 = d2 = None
        out["real_pair"] = {"skipped": repr(e)}
    if d1 is not None and (args.width, args.height) == (640, 480):
        kintr = synth.Intrinsics()            # the fixtures' intrinsics: 525 / 525 / 319.5 / 235.5 / 1000 (src/convert2PCD.cpp:19-23)
        def real_legs(ke):
          kparams = capi.default_params(kintr, iterations=args.iterations, max_batch=1, device=local_rank, **ke.kw())
          khandles = [capi.IcpHandle(kparams) for _ in handles]
          try:
            real = {}
            wide = synth.FramePair(-1, kintr, d1, d2, np.eye(4))
            r, leg = stream_leg(khandles, [Pool(torch, synth, None, 640, 480, 0.0, pairs=[wide]) for _ in khandles], 512)
            leg.update({"n_src": r["n_src"], "n_tgt": r["n_tgt"], "inliers": r["inliers"], "status": r["status"], "norm": r["norm"]})
            real["dep1_to_dep2_wide_baseline"] = leg
            Ti = synth.pose_from_seed(77, 2.0, 0.03)
            for name, d in (("dep1_to_dep1_perturbed", d1), ("dep2_to_dep2_perturbed", d2)):
                same = synth.FramePair(-1, kintr, d, d, np.eye(4))
                r, leg = stream_leg(khandles, [Pool(torch, synth, None, 640, 480, 0.0, pairs=[same]) for _ in khandles], 512, T_init=Ti.reshape(1, 16))
                rot = float(np.arccos(min(1.0, max(-1.0, (np.trace(np.array(r["T_raw"]).reshape(4, 4)[:3, :3]) - 1.0) / 2.0))))
                leg.update({"n_src": r["n_src"], "n_tgt": r["n_tgt"], "inliers": r["inliers"], "status": r["status"], "residual_rot_rad": rot,
                            "residual_trans_m": float(np.linalg.norm(np.array(r["T_raw"]).reshape(4, 4)[:3, 3]))})
                real[name] = leg
            real["estimator"] = ke.name
            real["note"] = ("the reference's 640x480 Kinect depth images (tests/golden/kinect), H2D of both images inside every alignment, "
                            f"{len(khandles)} in flight; perturbed legs start 2 deg / 3 cm away from the identity and must return to it")
            return real
          finally:
            for hh in khandles:
                hh.close()
        out["real_pair"] = real_legs(est)
        if est.lib != capi.EST_PLANE and "plane_normals" in out:
            out["plane_normals"]["real_pair"] = real_legs(Est(capi, "plane_gate", est.coarse))
    # ---- (ii-c2) round 5: ICP at the reference's ACTUAL operating point: readimage's cloud -- the frame's records
    # through PassThrough + VoxelGrid(0.03): 16,034 / 14,758 points for the reference's data/exp1 frames -- aligned as UNORGANIZED
    # clouds (height == 1 handle: the persistent list kernel, svd estimator), and the batched voxel grid
    if d1 is not None and (args.width, args.height) == (640, 480):
        try:
            out["voxel_icp"] = voxel_icp_leg(args, torch, capi, synth, local_rank, handles[0], d1, d2, "cpu_baseline" in out)
        except Exception as e:      # noqa: BLE001 -- the leg is optional
            out["voxel_icp"] = {"skipped": repr(e)}
    # ---- (ii-d) the same stream with TWO pairs per launch sequence (still four sequences in flight): what the per-launch fixed
    # costs are worth -- a settled launch costs 22 us before any lane searches and a second pair shares it.
    # Not the headline: config 2 is one pair per sequence.
    p2 = capi.default_params(intr, iterations=args.iterations, max_batch=2, device=local_rank, **est.kw())
    h2 = [capi.IcpHandle(p2) for _ in handles]
    st2 = Streamer(h2, pools, 2)
    st2.run(16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k2 = 384
    st2.run(k2)
    torch.cuda.synchronize()
    v2 = 2 * k2 * args.iterations / (time.perf_counter() - t0)
    out["two_pairs_per_launch"] = {"value": v2, "ratio_to_headline": v2 / out["value"], "alignments": k2, "pairs_per_launch": 2,
                                   "note": f"same streaming regime, two pairs per launch sequence, {len(h2)} sequences in flight"}
    for hh in h2:
        hh.close()
    # ---- (iii) BASELINE config 3: 64 pairs per launch sequence
    P3 = 64
    pool3 = Pool(torch, synth, [args.seed0 + k for k in range(P3)], args.width, args.height, args.noise_sigma, (args.hole_block, args.hole_prob))
    p3 = capi.default_params(intr, iterations=args.iterations, max_batch=P3, device=local_rank, **est.kw())
    h3 = [capi.IcpHandle(p3) for _ in range(3)]
    st3 = Streamer(h3, [pool3, pool3, pool3], P3)
    st3.run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k3 = 36
    res3 = st3.run(k3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof3 = profiled_pass(h3[0], pool3, P3, 3, est)
    c3 = {"workload": f"BASELINE config 3: {P3} pairs (seeds {args.seed0}..{args.seed0 + P3 - 1}) per launch sequence, {k3} alignments timed, "
                      "3 in flight, H2D of all 128 depth images inside every alignment",
          "value": k3 * P3 * args.iterations / dt, "unit": "ICP iterations/s", "ms_per_alignment": 1e3 * dt / k3,
          "kernel_only_value": P3 * args.iterations / (prof3["total_ms"] * 1e-3),
          "roofline": tiles_roofline(prof3, args.iterations, "k_nn_tiles_acc_640x480_P64",
                                     "; with 64 pairs per launch the kernel runs at ~97 % VALU issue (throughput build)"),
          "status_ok": int(sum(1 for r in res3 if r["status"] == 0))}
    if "cpu_baseline" in out:
        c3["cpu_baseline"] = dict(out["cpu_baseline"], sample=out["cpu_baseline"]["sample"] + " -- pairs are independent: the CPU rate per "
                                  "pair is the same for a batch (one pair of the batch timed)")
    out["config3"] = c3
    for h in h3:
        h.close()
    # ---- (iv) BASELINE config 5 on one GPU: 1280x960 dense
    d = dense_leg(args, torch, dist, capi, synth, 1, 0, local_rank, None, 1280, 960, 24, 3, want_cpu=("cpu_baseline" in out))
    c5 = {"workload": "BASELINE config 5 (1-GPU leg): one 1280x960 pair per step (seeds 2000..2003), slam3d_icp_dense_run, H2D of both depth "
                      "images inside every step, 24 steps timed",
          "value": 24 * args.iterations / d["elapsed"], "unit": "ICP iterations/s", "ms_per_step": 1e3 * d["elapsed"] / 24,
          "roofline": tiles_roofline(d["prof"], args.iterations, "k_nn_tiles_acc_1280x960"),
          "n_src": d["res"]["n_src"], "n_tgt": d["res"]["n_tgt"], "target_preprocessing": d.get("target_preprocessing")}
    if "cpu" in d:
        c5["cpu_baseline"], c5["parity_vs_oracle"] = d["cpu"], d["parity"]
    out["config5"] = c5
    d["handle"].close()


if __name__ == "__main__":
    main()

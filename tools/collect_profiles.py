#!/usr/bin/env python3
"""Turns gpurun_out/refresh_<tag>/ (tools/refresh_profiles.sh) into the committed files under profiles/:
  <tag>_bench.json            the default bench line
  <tag>_kernel_stats.md       rocprofv3 --kernel-trace --stats summary + the PMC passes per dispatch
  <tag>_traffic.json          HBM traffic per launch of the dominant kernel (three configurations) + the SHA of the kernel
                              source it was measured on (bench.py refuses a stale profile)
  <tag>_pipelined_trace.md    kernel trace of the 3-in-flight regime: concurrency, per-kernel duration under overlap
  <tag>_voxel_* / seg         rows f-1 / f-2
usage: tools/collect_profiles.py r02"""
import glob
import hashlib
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", f"refresh_{tag}")
# the rocpd databases are too large to travel back from the GPU box: refresh_profiles.sh runs this script THERE with a
# destination under gpurun_out/ (which is merged back); copy those small files into profiles/ afterwards
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def last_json(path):
    line = [ln for ln in open(path) if ln.startswith("{")][-1]
    return json.loads(line)


def db_of(sub):
    g = glob.glob(os.path.join(src, sub, "**", "*.db"), recursive=True)
    return g[0] if g else None


bench = last_json(os.path.join(src, "bench.json"))
json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
for name in ("voxel_bench", "seg64_bench", "seg1_bench"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p):
        try:
            json.dump(last_json(p), open(os.path.join(dst, f"{tag}_{name}.json"), "w"), indent=1)
        except Exception:
            pass

# ---------------------------------------------------------------------------------------- kernel stats + PMC
md = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), db_of("stats"),
                     "--title", f"{tag}: rocprofv3 --kernel-trace --stats",
                     "--cmd", "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline   "
                              "(timed stream 3 in flight + latency leg + event-profiled pass + the brute-force legs k_nn_mfma / k_nn_valu)"],
                    capture_output=True, text=True, check=True).stdout


def pmc_rows(prefix):
    rows = {}
    for db in sorted(glob.glob(os.path.join(src, prefix + "_*", "**", "*.db"), recursive=True)):
        c = sqlite3.connect(db)
        q = ("select name, counter_name, avg(v), count(*) from (select name, dispatch_id, counter_name, sum(counter_value) v from pmc_events "
             "where name like '%s3d::%' group by dispatch_id, counter_name) group by name, counter_name")
        for name, ctr, v, n in c.execute(q):
            rows[(name.split("(")[0], ctr)] = (v, n)
        if "SQ_WAVE_CYCLES" in db:
            r = c.execute("select avg(duration) from kernels where name like '%k_nn_tiles_acc%'").fetchone()
            rows[("_dur", "k_nn_tiles_acc")] = (r[0], 0)
    return rows


CFG = {"P1": ("k_nn_tiles_acc_640x480_P1", "single 640x480 pairs, one at a time (cooperative build)",
              "bench.py --steps 1 --warmup 1 --pairs-per-step 24 --no-pipeline --timed-only"),
       "P64": ("k_nn_tiles_acc_640x480_P64", "64 pairs of 640x480 per launch (throughput build)",
               "bench.py --steps 1 --warmup 1 --pairs 64 --pairs-per-step 64 --pool 64 --no-pipeline --timed-only"),
       "D": ("k_nn_tiles_acc_1280x960", "one 1280x960 pair (config 5, 1-GPU leg)", "bench.py --mode dense --width 1280 --height 960 --steps 2 --warmup 1")}
traffic = {"_comment": ("HBM traffic per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, KiB per "
                        "dispatch summed over the XCD rows, averaged over the dispatches).  FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM "
                        "section) prescribes for wide coalesced reads on gfx950.  bench.py uses an entry only while kernel_src_sha16 matches "
                        "slam3d_gx_amd/csrc/icp_kernels.hpp."),
           "kernel_src_sha16": hashlib.sha256(open(os.path.join(ROOT, "slam3d_gx_amd", "csrc", "icp_kernels.hpp"), "rb").read()).hexdigest()[:16]}
for key, (tname, what, cmd) in CFG.items():
    rows = pmc_rows("pmc_" + key)
    if not rows:
        continue
    md += (f"\n## PMC passes: {what}\n\n`rocprofv3 --kernel-trace --pmc <counters> -- python {cmd}` (one run per counter set).  Per dispatch, summed over "
           "the XCD/SE rows of the dispatch, averaged over the dispatches.  FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE under-reports wide "
           "coalesced reads by 2x (MI355X_MICROARCH.md, HBM section); SQ_*_CYCLES count quad-cycles.\n\n| kernel | counter | per dispatch | dispatches |\n|---|---|---|---|\n")
    for (name, ctr), (v, n) in sorted(rows.items()):
        if name != "_dur":
            md += f"| `{name}` | {ctr} | {v:.1f} | {n} |\n"
    cands = sorted({n for (n, _c) in rows if "k_nn_tiles_acc" in n})
    if not cands:
        continue
    k = cands[0]
    if (k, "FETCH_SIZE") in rows and (k, "WRITE_SIZE") in rows:
        f, w = rows[(k, "FETCH_SIZE")][0], rows[(k, "WRITE_SIZE")][0]
        e = {"kernel_instance": k, "fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024),
             "hbm_bytes_per_launch_uncorrected": int((f + w) * 1024), "dispatches": rows[(k, "FETCH_SIZE")][1]}
        sq = {c: rows[(k, c)][0] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                                            "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY") if (k, c) in rows}
        if sq:
            e["sq_counters_per_launch"] = {kk: round(v, 1) for kk, v in sq.items()}
            if "SQ_INSTS_VALU" in sq:
                e["valu_instructions_per_wave"] = round(sq["SQ_INSTS_VALU"] / max(sq.get("SQ_WAVES", 1), 1), 1)
                # 4 cycles per wave64 VALU instruction on a 16-lane SIMD, 1024 SIMDs, ~2.3 GHz under load
                e["valu_issue_floor_us"] = round(sq["SQ_INSTS_VALU"] * 4 / 1024 / 2.3e3, 2)
        if ("_dur", "k_nn_tiles_acc") in rows and rows[("_dur", "k_nn_tiles_acc")][0]:
            e["launch_ns_in_pmc_run"] = round(rows[("_dur", "k_nn_tiles_acc")][0], 1)
        traffic[tname] = e
open(os.path.join(dst, f"{tag}_kernel_stats.md"), "w").write(md)
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)


# ---------------------------------------------------------------------------------------- pipelined trace
def trace_summary(db, title, bench_json, note):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, queue_id from kernels where name like '%s3d::%' order by start").fetchall()
    if not rows:
        return ""
    b = last_json(bench_json)
    # the timed region = the last (steps x alignments) runs: every run starts with k_pair_init
    names = [r[0].split("(")[0] for r in rows]
    inits = [i for i, n in enumerate(names) if "k_pair_init" in n]
    n_align = b["steps"] * b["config"]["alignments_per_step"]
    first = inits[-n_align] if len(inits) >= n_align else 0
    # back up to the preprocessing kernels of that alignment (they precede its k_pair_init on the same queue)
    rows = rows[max(0, first - 8):]
    names = names[max(0, first - 8):]
    t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
    ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
    conc, last, level = {}, t0, 0
    for t, d in ev:
        conc[level] = conc.get(level, 0) + (t - last)
        last, level = t, level + d
    window = t1 - t0
    per = {}
    for n, r in zip(names, rows):
        a = per.setdefault(n, [0, 0.0])
        a[0] += 1; a[1] += r[2] - r[1]
    busy = sum(v for k, v in conc.items() if k > 0)
    total_kernel_ns = sum(a[1] for a in per.values())
    out = f"# {title}\n\n{note}\n\n"
    out += (f"window {window / 1e6:.3f} ms, {n_align} alignments of {b['config']['pairs_per_launch']} pair(s) x {b['config']['iterations']} iterations "
            f"-> {window / 1e3 / n_align:.1f} us per alignment in the traced run (bench line of the same run: {b['ms_per_step'] / b['config']['alignments_per_step'] * 1e3:.1f} us, "
            f"{b['value']:.0f} it/s; rocprofv3 tracing adds to both)\n\n")
    out += "| kernels in flight | time | share of the window |\n|---|---|---|\n"
    for k in sorted(conc):
        out += f"| {k} | {conc[k] / 1e3:.1f} us | {100 * conc[k] / window:.1f} % |\n"
    out += (f"\nGPU busy (>= 1 kernel) {100 * busy / window:.1f} % of the window; sum of kernel durations {total_kernel_ns / 1e6:.3f} ms = "
            f"{total_kernel_ns / window:.2f} x the window (= mean concurrency).\n\n")
    out += "| kernel | dispatches | avg duration us | total ms | per alignment us |\n|---|---|---|---|---|\n"
    for n, a in sorted(per.items(), key=lambda kv: -kv[1][1]):
        out += f"| `{n}` | {a[0]} | {a[1] / a[0] / 1e3:.2f} | {a[1] / 1e6:.3f} | {a[1] / 1e3 / n_align:.1f} |\n"
    nn = [a for n, a in per.items() if "k_nn_tiles_acc" in n]
    if nn:
        d_nn = nn[0][1] / nn[0][0] / 1e3
        out += (f"\nCheck (VERDICT r1 item 1): {b['config']['iterations']} x (dominant-kernel duration under overlap {d_nn:.2f} us) / (mean concurrency "
                f"{total_kernel_ns / window:.2f}) = {b['config']['iterations'] * d_nn / (total_kernel_ns / window):.1f} us <= {window / 1e3 / n_align:.1f} us per alignment "
                f"(the rest of the per-alignment time is the other kernels and the gaps).\n")
    return out


text = ""
if db_of("pipe"):
    text += trace_summary(db_of("pipe"), f"{tag}: kernel trace of the pipelined regime (bench.py's default: 8 alignments in flight)",
                          os.path.join(src, "pipe_bench.json"),
                          "`rocprofv3 --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline --no-bruteforce --timed-only`: "
                          "eight handles, one HIP stream each, every alignment uploads both depth images and rebuilds normals + tiles.")
if db_of("solo"):
    text += "\n\n" + trace_summary(db_of("solo"), f"{tag}: the same stream, one alignment at a time (--no-pipeline)",
                                   os.path.join(src, "solo_bench.json"),
                                   "`... bench.py --steps 1 --warmup 1 --pairs-per-step 96 --no-pipeline --timed-only`: kernel durations without overlap.")
if text:
    open(os.path.join(dst, f"{tag}_pipelined_trace.md"), "w").write(text)
if db_of("vox"):
    v = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), db_of("vox"), "--title", f"{tag}: row f-1 voxel grid, rocprofv3 --kernel-trace --stats",
                        "--cmd", "rocprofv3 --kernel-trace --stats -- python bench.py --mode voxel --steps 50 --warmup 5 --no-cpu-baseline"],
                       capture_output=True, text=True, check=True).stdout
    open(os.path.join(dst, f"{tag}_voxel_kernel_stats.md"), "w").write(v)
print(json.dumps({k2: bench[k2] for k2 in ("value", "ms_per_step")}), "written", tag)

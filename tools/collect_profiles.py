#!/usr/bin/env python3
"""Turns gpurun_out/refresh/ (tools/refresh_profiles.sh) into the committed files under profiles/:
<tag>_bench.json, <tag>_kernel_stats.md (kernel stats + PMC per dispatch) and r01_traffic.json."""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
src = os.path.join(ROOT, "gpurun_out", "refresh")
dst = os.path.join(ROOT, "profiles")

line = [ln for ln in open(os.path.join(src, f"{tag}_bench.json")) if ln.startswith("{")][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)

md = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), glob.glob(os.path.join(src, "stats", "*.db"))[0],
                     "--title", f"{tag}: rocprofv3 --kernel-trace --stats",
                     "--cmd", "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline   (includes the brute-force legs: k_nn_mfma, k_nn_valu)"],
                    capture_output=True, text=True, check=True).stdout
rows = {}
for db in sorted(glob.glob(os.path.join(src, "pmc_*", "*.db"))):
    c = sqlite3.connect(db)
    q = ("select name, counter_name, avg(v), count(*) from (select name, dispatch_id, counter_name, sum(counter_value) v from pmc_events "
         "where name like '%s3d::%' group by dispatch_id, counter_name) group by name, counter_name")
    for name, ctr, v, n in c.execute(q):
        rows[(name.split("(")[0], ctr)] = (v, n)
md += ("\n## PMC passes (separate runs, `rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 2 --warmup 1 "
       "--no-cpu-baseline --no-bruteforce`)\n\nPer dispatch, summed over the XCD/SE rows of the dispatch, averaged over the dispatches of the run. "
       "FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section); "
       "SQ_*_CYCLES count quad-cycles.\n\n| kernel | counter | per dispatch | dispatches |\n|---|---|---|---|\n")
for (name, ctr), (v, n) in sorted(rows.items()):
    md += f"| `{name}` | {ctr} | {v:.1f} | {n} |\n"
open(os.path.join(dst, f"{tag}_kernel_stats.md"), "w").write(md)

cands = sorted({n for (n, _c) in rows if "k_nn_tiles_acc" in n})       # template instance, e.g. "void s3d::k_nn_tiles_acc<3, 8, true>"
k = cands[0] if cands else "s3d::k_nn_tiles_acc"
if (k, "FETCH_SIZE") in rows and (k, "WRITE_SIZE") in rows:
    f, w = rows[(k, "FETCH_SIZE")][0], rows[(k, "WRITE_SIZE")][0]
    out = {"_comment": ("HBM traffic of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, KiB per "
                        f"dispatch, averaged over {rows[(k, 'FETCH_SIZE')][1]} dispatches of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                        "--no-bruteforce`). FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on "
                        f"gfx950. Kernel instance: {k}. See profiles/{tag}_kernel_stats.md."),
           "k_nn_tiles_acc": {"fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                              "hbm_bytes_per_launch_uncorrected": int((f + w) * 1024),
                              "algorithmic_bytes_per_launch": bench.get("roofline", {}).get("algorithmic_bytes_per_launch")}}
    sq = {c: rows[(k, c)][0] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                                        "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY") if (k, c) in rows}
    if "SQ_ACTIVE_INST_VALU" in sq and (k, "SQ_BUSY_CYCLES") in rows:
        # quad-cycle counters: VALU-active quad-cycles summed over the 1024 SIMDs / (launch duration in quad-cycles x 1024)
        dur_ns = None
        try:
            c = sqlite3.connect(glob.glob(os.path.join(src, "pmc_SQ_WAVE_CYCLES", "*.db"))[0])
            dur_ns = c.execute("select avg(duration) from kernels where name like '%k_nn_tiles_acc%'").fetchone()[0]
        except Exception:
            pass
        out["k_nn_tiles_acc"]["sq_counters_per_launch"] = {kk: round(v, 1) for kk, v in sq.items()}
        if dur_ns:
            out["k_nn_tiles_acc"]["launch_ns_in_pmc_run"] = round(dur_ns, 1)
            out["k_nn_tiles_acc"]["valu_instructions_per_wave"] = round(sq.get("SQ_INSTS_VALU", 0) / max(sq.get("SQ_WAVES", 1), 1), 1)
            out["k_nn_tiles_acc"]["waves_per_launch_note"] = ("SQ_WAVES counts every launched wave; the single-pair grid carries 20 % slack "
                                                              "(waves without a tile that only help with shared work items), so per "
                                                              "tile-owning wave the count is SQ_INSTS_VALU / ntiles (4800 tiles at 640x480)")
            # 4 cycles per wave64 VALU instruction on a 16-lane SIMD, 1024 SIMDs, ~2.3 GHz
            out["k_nn_tiles_acc"]["valu_issue_floor_us"] = round(sq.get("SQ_INSTS_VALU", 0) * 4 / 1024 / 2.3e3, 2)
    json.dump(out, open(os.path.join(dst, "r01_traffic.json"), "w"), indent=1)
print(json.dumps({k2: bench[k2] for k2 in ("value", "ms_per_step")}), "written", tag)

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
         "where name like 's3d::%' group by dispatch_id, counter_name) group by name, counter_name")
    for name, ctr, v, n in c.execute(q):
        rows[(name.split("(")[0], ctr)] = (v, n)
md += ("\n## PMC passes (separate runs, `rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 2 --warmup 1 "
       "--no-cpu-baseline --no-bruteforce`)\n\nPer dispatch, summed over the XCD/SE rows of the dispatch, averaged over the dispatches of the run. "
       "FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section); "
       "SQ_*_CYCLES count quad-cycles.\n\n| kernel | counter | per dispatch | dispatches |\n|---|---|---|---|\n")
for (name, ctr), (v, n) in sorted(rows.items()):
    md += f"| `{name}` | {ctr} | {v:.1f} | {n} |\n"
open(os.path.join(dst, f"{tag}_kernel_stats.md"), "w").write(md)

k = "s3d::k_nn_tiles_acc"
if (k, "FETCH_SIZE") in rows and (k, "WRITE_SIZE") in rows:
    f, w = rows[(k, "FETCH_SIZE")][0], rows[(k, "WRITE_SIZE")][0]
    out = {"_comment": ("HBM traffic of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, KiB per "
                        f"dispatch, averaged over {rows[(k, 'FETCH_SIZE')][1]} dispatches of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                        "--no-bruteforce`). FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on "
                        f"gfx950. See profiles/{tag}_kernel_stats.md."),
           "k_nn_tiles_acc": {"fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                              "hbm_bytes_per_launch_uncorrected": int((f + w) * 1024),
                              "algorithmic_bytes_per_launch": bench.get("roofline", {}).get("algorithmic_bytes_per_launch")}}
    json.dump(out, open(os.path.join(dst, "r01_traffic.json"), "w"), indent=1)
print(json.dumps({k2: bench[k2] for k2 in ("value", "ms_per_step")}), "written", tag)

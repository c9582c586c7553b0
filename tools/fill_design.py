#!/usr/bin/env python3
"""Fills the @@PLACEHOLDERS@@ of DESIGN.md section 7 / 9 from a bench line (profiles/rNN_bench.json) so that the table is a
copy of the measured file, not a transcription.  usage: tools/fill_design.py profiles/r04_bench.json [n_gpu_tests] [normals_us]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(sys.argv[1]).read()
try:
    d = json.loads(txt)
except Exception:
    d = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
k = lambda v: f"{v / 1e3:.1f} k"
per = d["nn_ms_per_iteration"]
rp = d.get("real_pair", {})
rep = {
    "HEADLINE": f"{d['value']:,.0f} ({1e3 * 20 / d['value']:.3f} ms per pair)",
    "LAT": f"{d['single_step_latency_ms']:.3f}",
    "PERIT": " / ".join(f"{1e3 * x:.0f}" for x in per[:4]) + f", {1e3 * min(per[4:8]):.0f}–{1e3 * max(per[4:8]):.0f}, {1e3 * min(per[8:]):.0f}–{1e3 * max(per[8:]):.0f}",
    "BMD": f"{d['baseline_md_workload']['value']:,.0f} ({d['baseline_md_workload']['ratio_to_headline']:.2f}×; n_tgt {d['baseline_md_workload']['n_tgt']:,})" if "baseline_md_workload" in d else "n/a",
    "REAL": " / ".join(k(rp[n]["value"]) for n in ("dep1_to_dep2_wide_baseline", "dep1_to_dep1_perturbed", "dep2_to_dep2_perturbed") if n in rp),
    "C3": f"{d['config3']['value']:,.0f}" if "config3" in d else "n/a",
    "C5": f"{d['config5']['value']:,.0f}" if "config5" in d else "n/a",
    "NTESTS": sys.argv[2] if len(sys.argv) > 2 else "?",
    "NRM": (sys.argv[3] + " µs (71 VGPRs)") if len(sys.argv) > 3 else "?",
}
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for key, v in rep.items():
    s = s.replace("@@" + key + "@@", v)
open(p, "w").write(s)
print({key: v for key, v in rep.items()})
left = re.findall(r"@@\w+@@", s)
if left:
    print("unfilled:", left)

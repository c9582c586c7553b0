#!/usr/bin/env python3
"""Developer tool: register / LDS / scratch budget of every kernel of the library, from the code-object metadata
(device-only compile of csrc/icp_capi.hip with the library's flags).  usage: python tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam3d_gx_amd import build as B

flt = sys.argv[1] if len(sys.argv) > 1 else ""
flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
extra = [a for a in sys.argv[2:]]
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "dev.s")
    subprocess.check_call([B.hipcc()] + flags + extra + ["--cuda-device-only", "-S", os.path.join(B.CSRC, "icp_capi.hip"), "-o", asm])
    notes = open(asm).read()
    if os.environ.get("KEEP_ASM"):
        open(os.environ["KEEP_ASM"], "w").write(notes)
    demangle = "/usr/bin/c++filt"
recs, cur = [], None
for line in notes.splitlines():
    if re.match(r"  - \.\w+:", line):          # a new kernel record of amdhsa.kernels (keys are alphabetical: .name comes late)
        cur = {}
        recs.append(cur)
    m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", line)
    if not m or cur is None:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "name" and v.startswith("_Z"):
        cur["name"] = v
    elif k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
               "group_segment_fixed_size", "agpr_count"):
        cur[k] = int(v)
kern = {r["name"]: r for r in recs if "name" in r}
names = list(kern)
dem = subprocess.check_output([demangle] + names, text=True).splitlines() if names else []
print(f"{'kernel':70s} vgpr agpr sgpr vspill sspill scratch   lds")
for n, d in zip(names, dem):
    if flt and flt not in d:
        continue
    r = kern[n]
    short = re.sub(r"\(.*", "", d).replace("s3d::", "").replace("void ", "")
    print(f"{short[:70]:70s} {r.get('vgpr_count', 0):4d} {r.get('agpr_count', 0):4d} {r.get('sgpr_count', 0):4d} {r.get('vgpr_spill_count', 0):6d} "
          f"{r.get('sgpr_spill_count', 0):6d} {r.get('private_segment_fixed_size', 0):7d} {r.get('group_segment_fixed_size', 0):5d}")

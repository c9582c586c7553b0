#!/bin/bash
# Runs ON THE GPU BOX: A/B of library builds (tools/variants/<name>.so) on the default bench stream.
# usage: bash tools/ab_lib.sh <tag> "<name1> <name2> ..." [extra bench args]
set -u
TAG=${1:-ab}; NAMES=${2:-"base"}; shift 2 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
Q="--no-extra-configs --no-cpu-baseline --no-bruteforce --steps 10 --warmup 3"
i=0
for m in $NAMES; do
    i=$((i+1))
    SLAM3D_LIB=$R/tools/variants/$m.so timeout 600 python bench.py $Q --legs-file $OUT/$m.$i.json "$@" > $OUT/$m.$i.out 2> $OUT/$m.$i.err
    python - <<PY
import json
d=json.load(open("$OUT/$m.$i.json"))
o=d.get("overlap",{})
print("%-10s value %.0f  latency %.3f ms  kernels/align %.3f ms (nn %.3f)  launch %.2f us | stamped: value %.0f resident %.2f nn %.1f us" % ("$m",
  d["value"], d["single_step_latency_ms"], d["kernel_ms_per_alignment"]["total"], d["kernel_ms_per_alignment"]["nn"], 1e3*d["roofline"]["launch_ms"],
  o.get("value_while_stamping",0), o.get("mean_resident_nn_kernels",0), o.get("nn_launch_us_overlapped",{}).get("mean",0)))
print("   nn us per iteration:", [round(1e3*x,1) for x in d["nn_ms_per_iteration"]])
PY
done

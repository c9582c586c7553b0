#!/bin/bash
# Runs ON THE GPU BOX: A/B of one developer knob on the default bench stream.
# usage: bash tools/ab_env.sh <tag> <ENV_VAR> "<v1> <v2> ..." [extra bench args]   e.g.  bash tools/ab_env.sh ab SLAM3D_CERT "1 0 1 0"
set -u
TAG=${1:-ab}; VAR=${2:-SLAM3D_CERT}; VALS=${3:-"1 0"}; shift 3 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
Q="--no-extra-configs --no-cpu-baseline --no-bruteforce --steps 10 --warmup 3"
for m in $VALS; do
    env $VAR=$m timeout 600 python bench.py $Q "$@" > $OUT/$VAR$m.json 2> $OUT/$VAR$m.err
    python - <<PY
import json
d=json.load(open("$OUT/$VAR$m.json"))
o=d.get("overlap",{})
print("$VAR=$m value %.0f  latency %.3f ms  kernels/align %.3f ms (nn %.3f)  launch %.2f us | stamped: value %.0f resident %.2f nn %.1f us" % (
  d["value"], d["single_step_latency_ms"], d["kernel_ms_per_alignment"]["total"], d["kernel_ms_per_alignment"]["nn"], 1e3*d["roofline"]["launch_ms"],
  o.get("value_while_stamping",0), o.get("mean_resident_nn_kernels",0), o.get("nn_launch_us_overlapped",{}).get("mean",0)))
print("   nn ms per iteration:", d["nn_ms_per_iteration"])
PY
done

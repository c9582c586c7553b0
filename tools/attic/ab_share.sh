#!/bin/bash
# Runs ON THE GPU BOX: A/B of the tail sharing (SLAM3D_TAIL_SHARE = 0 off, 1 while no other run is in flight, 2 always) on the default bench stream.
# polling, 2 head solve in every block) on the default bench stream.  usage: bash tools/ab_head.sh <tag> [extra bench args]
set -u
TAG=${1:-ab}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
Q="--no-extra-configs --no-cpu-baseline --no-bruteforce --steps 10 --warmup 3"
for m in ${MODES:-1 0 2 1 0}; do
    SLAM3D_TAIL_SHARE=$m timeout 600 python bench.py $Q "$@" > $OUT/share$m.json 2> $OUT/share$m.err
    python - <<PY
import json
d=json.load(open("$OUT/share$m.json"))
o=d.get("overlap",{})
print("share=$m value %.0f  latency %.3f ms  kernels/align %.3f ms (nn %.3f)  launch %.2f us | stamped: value %.0f resident %.2f nn %.1f us gap %.1f us" % (
  d["value"], d["single_step_latency_ms"], d["kernel_ms_per_alignment"]["total"], d["kernel_ms_per_alignment"]["nn"], 1e3*d["roofline"]["launch_ms"],
  o.get("value_while_stamping",0), o.get("mean_resident_nn_kernels",0), o.get("nn_launch_us_overlapped",{}).get("mean",0), o.get("nn_to_nn_gap_us",{}).get("mean",0)))
PY
done

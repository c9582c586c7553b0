#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of a short one-at-a-time stream (kernel durations without overlap).
# usage: bash tools/quick_stats.sh <tag> [bench args]
set -u
TAG=${1:-qs}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-extra-configs --no-cpu-baseline --no-bruteforce --timed-only --no-pipeline --steps 1 --warmup 1 --pairs-per-step 96"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/solo -o solo -- python $R/bench.py $Q "$@" > $OUT/solo_bench.json 2> $OUT/solo.err
DB=$(find $OUT/solo -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB --title "$TAG solo" --cmd "bench.py $Q $*" > $OUT/solo_stats.md 2>&1
rm -rf $OUT/solo
head -40 $OUT/solo_stats.md

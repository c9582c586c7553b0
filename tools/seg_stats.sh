#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the plane-segmentation launch sequence.  usage: bash tools/seg_stats.sh <tag> <frames>
set -u
TAG=${1:-segstats}; F=${2:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/seg -o seg -- python $R/bench.py --mode seg --pairs $F --steps 50 --warmup 5 --no-cpu-baseline > $OUT/seg_bench.json 2> $OUT/seg.err
DB=$(find $OUT/seg -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB --title "$TAG: plane segmentation, $F frame(s) per call" --cmd "bench.py --mode seg --pairs $F --steps 50 --warmup 5 --no-cpu-baseline" > $OUT/seg_stats.md 2>&1
rm -rf $OUT/seg
head -30 $OUT/seg_stats.md

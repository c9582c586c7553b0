#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the unorganized-cloud ICP (16 k x 15 k voxel clouds, NN_AUTO = bf16 scan).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/unorg
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
QU_AUTO_ONLY=1 timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/p -o p -- python $R/tools/quick_unorg.py > $OUT/run.log 2> $OUT/run.err
DB=$(find $OUT/p -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB --title "unorganized ICP" --cmd "tools/quick_unorg.py (QU_AUTO_ONLY=1)" > $OUT/stats.md 2>&1
rm -rf $OUT/p
cat $OUT/run.log; head -30 $OUT/stats.md | cut -c1-200

#!/bin/bash
# Runs ON THE GPU BOX: parity of the voxel grid, then its bench and per-kernel times.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/exp_voxel; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_voxel.py -x -q -m gpu > $OUT/parity.log 2>&1; echo "parity rc=$?"; tail -3 $OUT/parity.log
timeout 300 python bench.py --mode voxel --steps 200 --warmup 20 --no-cpu-baseline > $OUT/voxel_bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/voxel_bench.json')); print('value', d['value'], 'ms_per_step', d['ms_per_step'], d['config'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/vox -o vox -- python $R/bench.py --mode voxel --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
DB=$(find $OUT/vox -name "*.db" | head -1); python $R/tools/rocpd_summary.py $DB | head -14; cat $OUT/vox/*kernel_stats.csv 2>/dev/null | head -12

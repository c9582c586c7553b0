#!/bin/bash
# Runs ON THE GPU BOX: A/B of library builds (tools/variants/<name>.so) on the single-frame segmentation and the plane-estimator stream.
# usage: bash tools/ab_seg.sh "<name1> <name2> ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do for m in $1; do
  a=$(SLAM3D_LIB=$R/tools/variants/$m.so python bench.py --mode seg --pairs 1 --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print('%.1f us' % (1e3*json.loads(sys.stdin.read())['ms_per_step']))")
  echo "$m seg1 $a"
done; done

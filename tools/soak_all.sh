#!/bin/bash
# Runs ON THE GPU BOX: the randomised parity soak on the current kernels -> gpurun_out/soak.log
# (cooperative build, throughput build, both brute-force modes, segmentation / voxel / transform side rows)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; L=gpurun_out/soak.log; : > $L
timeout 1500 python tools/soak_parity.py ${1:-400} 11 0 | tail -n 3 >> $L
SLAM3D_DENSE_BATCH=1 timeout 1200 python tools/soak_parity.py ${2:-300} 12 0 | tail -n 3 >> $L
timeout 600 python tools/soak_parity.py 80 13 2 | tail -n 3 >> $L
timeout 600 python tools/soak_parity.py 60 14 1 | tail -n 3 >> $L
timeout 600 python tools/soak_side.py 200 15 | tail -n 3 >> $L
cat $L

#!/bin/bash
# Runs ON THE GPU BOX: randomised parity soak of the full-scan kernels (bf16-split MFMA, f32 MFMA, VALU, VALU + filter) and of
# the side rows (voxel grid with first-writer slots, segmentation, transform) -> gpurun_out/soak_brute.log
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; L=gpurun_out/soak_brute.log; : > $L
echo "== NN_BRUTE_MFMA, bf16-split form (default)" >> $L; timeout 900 python tools/soak_parity.py ${1:-200} 21 2 | tail -n 2 >> $L
echo "== NN_BRUTE_MFMA, f32 form (SLAM3D_MFMA_BF16=0)" >> $L; SLAM3D_MFMA_BF16=0 timeout 900 python tools/soak_parity.py ${1:-200} 22 2 | tail -n 2 >> $L
echo "== NN_BRUTE_VALU, canonical (default)" >> $L; timeout 900 python tools/soak_parity.py ${1:-200} 23 1 | tail -n 2 >> $L
echo "== NN_BRUTE_VALU, expanded-form filter (SLAM3D_VALU_FILTER=1)" >> $L; SLAM3D_VALU_FILTER=1 timeout 900 python tools/soak_parity.py ${1:-200} 24 1 | tail -n 2 >> $L
echo "== side rows (voxel / segmentation / transform)" >> $L; timeout 600 python tools/soak_side.py 200 25 | tail -n 2 >> $L
cat $L

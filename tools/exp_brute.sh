#!/bin/bash
# Runs ON THE GPU BOX: parity of the brute modes with the in-tree build, then launch time of the full-scan kernels per configuration.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/exp_brute; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gates.py -x -q -m gpu -k "nn_mode or brute or valu or mfma" > $OUT/parity.log 2>&1; echo "parity rc=$?" ; tail -3 $OUT/parity.log
SLAM3D_MFMA_BF16=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nn_mode" > $OUT/parity_f32.log 2>&1; echo "parity (f32 mfma) rc=$?" ; tail -1 $OUT/parity_f32.log
for v in "" $(ls tools/variants/*.so 2>/dev/null); do
  echo "== ${v:-in-tree}"
  env ${v:+SLAM3D_LIB=$R/$v} timeout 300 python tools/quick_brute.py "SLAM3D_MFMA_SPLIT=20" 2>&1 | tail -1
done | tee $OUT/brute.log

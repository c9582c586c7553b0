#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the GPU test suite, the per-wave NN statistics and the default bench line.
# usage: bash tools/gpu_round.sh <tag> [pytest-args...]      outputs under gpurun_out/<tag>/
set -u
TAG=${1:-round}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -n 30 $OUT/pytest.log
timeout 300 python tools/nn_debug.py 20 > $OUT/nn_debug.log 2>&1; tail -n 30 $OUT/nn_debug.log
timeout 900 python bench.py --legs-file $OUT/bench.json > $OUT/bench_stdout.txt 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench.err; tail -n 1 $OUT/bench_stdout.txt | wc -c; tail -n 1 $OUT/bench_stdout.txt

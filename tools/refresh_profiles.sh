#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): default bench line, rocprofv3 kernel stats and the PMC passes the roofline
# object cites; everything lands under gpurun_out/refresh/ (copy into profiles/ afterwards with
# tools/collect_profiles.py).  usage: bash tools/refresh_profiles.sh <tag>
set -u
TAG=${1:-r01_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
    n=$(echo $set | cut -d" " -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$n -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bruteforce > /dev/null 2>&1
done
ls -R $OUT | head -40

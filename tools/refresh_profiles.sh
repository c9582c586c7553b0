#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the default bench line, rocprofv3 kernel stats, the kernel trace of the pipelined
# regime and the PMC passes the roofline objects cite; everything lands under gpurun_out/refresh_<tag>/ (turn it into
# profiles/<tag>_* afterwards with tools/collect_profiles.py <tag>).  usage: bash tools/refresh_profiles.sh r03
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/refresh_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-extra-configs --no-cpu-baseline --no-bruteforce"
timeout 1200 python $R/bench.py --legs-file $OUT/bench.json > $OUT/bench_stdout.txt 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace -d $OUT/pipe -o pipe -- python $R/bench.py --steps 3 --warmup 1 $Q --timed-only > $OUT/pipe_bench.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace -d $OUT/solo -o solo -- python $R/bench.py --steps 1 --warmup 1 --pairs-per-step 96 --no-pipeline $Q --timed-only > $OUT/solo_bench.json 2> /dev/null
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
    n=$(echo $set | cut -d" " -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_P1_$n -o pmc -- python $R/bench.py --steps 1 --warmup 1 --pairs-per-step 24 --no-pipeline $Q --timed-only > /dev/null 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_P64_$n -o pmc -- python $R/bench.py --steps 1 --warmup 1 --pairs 64 --pairs-per-step 64 --pool 64 --no-pipeline $Q --timed-only > /dev/null 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_D_$n -o pmc -- python $R/bench.py --mode dense --width 1280 --height 960 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/vox -o vox -- python $R/bench.py --mode voxel --steps 50 --warmup 5 --no-cpu-baseline > $OUT/voxel_bench.json 2> /dev/null
timeout 300 python $R/bench.py --mode voxel --steps 200 --warmup 20 > $OUT/voxel_bench.json 2> /dev/null
timeout 300 python $R/bench.py --mode seg --pairs 64 --steps 20 --warmup 3 > $OUT/seg64_bench.json 2> /dev/null
timeout 300 python $R/bench.py --mode seg --pairs 1 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/seg1_bench.json 2> /dev/null
# summarise on the box (the databases exceed what gpurun copies back), keep only the small files
python $R/tools/collect_profiles.py $TAG $R/gpurun_out/profiles_$TAG
# the default bench once more, now that the PMC traffic file belongs to THIS kernel source (bench.py ignores a stale one): its line and
# its complete object are the ones to commit
cp $R/gpurun_out/profiles_$TAG/${TAG}_traffic.json $R/profiles/${TAG}_traffic.json
timeout 1200 python $R/bench.py --legs-file $R/gpurun_out/profiles_$TAG/${TAG}_bench.json > $OUT/bench_stdout.txt 2> $OUT/bench.err
tail -n 1 $OUT/bench_stdout.txt > $R/gpurun_out/profiles_$TAG/${TAG}_bench_line.json
cp $OUT/*.json $OUT/bench.err $OUT/bench_stdout.txt $R/gpurun_out/profiles_$TAG/ 2>/dev/null
rm -rf $OUT
ls -la $R/gpurun_out/profiles_$TAG

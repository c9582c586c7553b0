#!/bin/bash
# Runs ON THE GPU BOX: SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE of k_nn_tiles_acc per ITERATION of a run (dispatch number mod iterations),
# single 640x480 pairs one at a time.  usage: bash tools/pmc_by_iteration.sh [tag]
set -u
TAG=${1:-pmcit}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-extra-configs --no-cpu-baseline --no-bruteforce --steps 1 --warmup 1 --pairs-per-step 24 --no-pipeline --timed-only"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $set | cut -d" " -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/$n -o pmc -- python $R/bench.py $Q > /dev/null 2>&1
done
python - "$OUT" <<'PY' | tee $OUT/by_iteration.md
import glob, os, sqlite3, sys
out = sys.argv[1]
tab = {}
for db in sorted(glob.glob(os.path.join(out, "*", "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    q = ("select dispatch_id, counter_name, sum(counter_value) from pmc_events where name like '%k_nn_tiles_acc%' group by dispatch_id, counter_name order by dispatch_id")
    per = {}
    for did, ctr, v in c.execute(q):
        per.setdefault(ctr, []).append(v)
    for ctr, vals in per.items():
        it = {}
        for k, v in enumerate(vals):
            it.setdefault(k % 20, []).append(v)
        tab[ctr] = [sum(it[i]) / len(it[i]) for i in range(20)]
print("| iteration | " + " | ".join(tab) + " |")
print("|---|" + "---|" * len(tab))
for i in range(20):
    print(f"| {i} | " + " | ".join(f"{tab[c][i]:.0f}" for c in tab) + " |")
PY
rm -rf $OUT/SQ_INSTS_VALU $OUT/FETCH_SIZE $OUT/WRITE_SIZE

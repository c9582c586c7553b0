#!/bin/bash
# Runs ON THE GPU BOX: PMC passes (one per counter set) over the full-scan kernels -> gpurun_out/pmc_brute/pmc_brute.md
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_brute; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
    n=$(echo $set | cut -d" " -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p_$n -o pmc -- python $R/tools/quick_brute.py "" "SLAM3D_MFMA_BF16=0" "SLAM3D_VALU_FILTER=1" > $OUT/run_$n.log 2>&1
done
python - <<PY > $OUT/pmc_brute.md
import glob, sqlite3
rows = {}
dur = {}
for db in sorted(glob.glob("$OUT/p_*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        q = ("select name, counter_name, avg(v), count(*) from (select name, dispatch_id, counter_name, sum(counter_value) v from pmc_events "
             "where name like '%k_nn_%' group by dispatch_id, counter_name) group by name, counter_name")
        for name, cn, v, n in c.execute(q):
            rows[(name.split("(")[0], cn)] = (v, n)
        for name, d, n in c.execute("select name, avg(duration), count(*) from kernels where name like '%k_nn_%' group by name"):
            dur[name.split("(")[0]] = (d / 1e3, n)
    except Exception as e:
        print("<!--", db, e, "-->")
print("# r05: PMC passes over the full-scan kernels (640x480 pair, tools/pmc_brute.sh; one rocprofv3 --pmc run per counter set, per dispatch, summed over XCDs / SEs)\n")
print("| kernel | counter | per dispatch | dispatches |\n|---|---|---|---|")
for (k, cn), (v, n) in sorted(rows.items()):
    print(f"| \`{k}\` | {cn} | {v:,.0f} | {n} |")
print("\n| kernel | avg duration under the counter runs (us) | dispatches |\n|---|---|---|")
for k, (d, n) in sorted(dur.items()):
    print(f"| \`{k}\` | {d:,.1f} | {n} |")
PY
cat $OUT/pmc_brute.md | head -60
rm -rf $OUT/p_*

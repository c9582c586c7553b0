#!/bin/bash
# Runs ON THE GPU BOX: quick A/B of environment knobs on the default bench line (timed region only).
# usage: bash tools/ab_sweep.sh VAR v1 v2 ...      (VAR=none: just one run)
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v timeout 300 python bench.py --steps 10 --warmup 3 --timed-only --no-extra-configs --no-cpu-baseline --no-bruteforce 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s', round(d['value']), 'ms/step', round(d['ms_per_step'],2))"
  env $VAR=$v timeout 300 python bench.py --steps 3 --warmup 1 --pairs-per-step 64 --no-pipeline --timed-only --no-extra-configs --no-cpu-baseline --no-bruteforce 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   unpipelined it/s', round(d['value']), 'ms/pair', round(d['ms_per_step']/64,4))"
done

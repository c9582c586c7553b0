#!/bin/bash
# Runs ON THE GPU BOX: the default bench line (timed region only) for several numbers of alignments in flight.
for f in "$@"; do
  echo "== in-flight=$f"
  timeout 300 python bench.py --steps 10 --warmup 3 --in-flight $f --timed-only --no-extra-configs --no-cpu-baseline --no-bruteforce 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s', round(d['value']), 'ms/step', round(d['ms_per_step'],2))"
done

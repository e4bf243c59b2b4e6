#!/bin/bash
# A/B of an environment variable in ONE gpurun call (same box): bash tools/ab_env.sh VAR VAL_A VAL_B  -> pairs/s + rocprof totals for each
export TMPDIR=/tmp
V=$1; shift
for val in "$@"; do
  echo "== $V=$val"
  for i in 1 2; do env $V=$val python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-corr-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('pairs/s', round(d['value'], 2), '1-in-flight', round(d.get('value_1_in_flight', 0), 2))"; done
done

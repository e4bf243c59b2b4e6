#!/bin/bash
# Round-6 A/B of the split3 switches in ONE gpurun call (same box), each configuration twice, ST_BENCH_CHILD=1 (no nested exact run):
#   bash tools/r6_ab.sh "ST_SPLIT3=0" "ST_SPLIT3=1 ST_S3_PE=0 ST_SPLIT3_SK2=0" "ST_SPLIT3=1 ST_S3_PE=0" "ST_SPLIT3=1"
export TMPDIR=/tmp
for cfg in "$@"; do
  for i in 1 2; do env ST_BENCH_CHILD=1 $cfg python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-corr-roofline --harness none 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$cfg: pairs/s', round(d['value'], 2), '1-in-flight', round(d.get('value_1_in_flight', 0), 2))"; done
done

#!/bin/bash
# the driver's invocation, five times on one box: spread of the short run
for i in 1 2 3 4 5; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-corr-roofline --harness none 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('run $i (20 steps): pairs/s', round(d['value'], 2), '1-in-flight', round(d.get('value_1_in_flight', 0), 2))"
done
python bench.py 2>gpurun_out/r5_bench.err | grep '^{' > gpurun_out/r5_bench.json
python -c "
import json; d = json.load(open('gpurun_out/r5_bench.json'))
print('default run: value', round(d['value'], 2), 'harness', d.get('harness_pairs_per_s'), 'batched', d.get('harness_batched_pairs_per_s'), 'frac', d['roofline']['frac'])"

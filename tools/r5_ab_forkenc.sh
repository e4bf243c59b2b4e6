#!/bin/bash
export TMPDIR=/tmp
ST_FORK_ENC=1 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "graph_replay or damped_eval_512" 2>&1 | tail -n 2
for f in 0 1 0 1; do
  ST_FORK_ENC=$f python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-corr-roofline --harness none 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('fork_enc $f: pairs/s', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 3), '1-in-flight', round(d.get('value_1_in_flight', 0), 2))"
done

#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py tests/test_properties_gpu.py -x -q -m gpu > gpurun_out/r5_ops_tests.log 2>&1; tail -n 3 gpurun_out/r5_ops_tests.log
python -m pytest tests/test_model_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r5_model_tests.log 2>&1; tail -n 3 gpurun_out/r5_model_tests.log
for cfg in "512 0" "256 1" "256 0" "512 1" "256 1" "512 0"; do
  set -- $cfg
  ST_PERSIST_SLOTS=$1 ST_PERSIST_CONV=$2 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-corr-roofline --harness none 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('slots $1 conv $2: pairs/s', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 3), '1-in-flight', round(d.get('value_1_in_flight', 0), 2))"
done

#!/bin/bash
# full GPU suite + the default bench line (GPU box, repo root)
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/r5_gpu_suite.log 2>&1; tail -n 5 gpurun_out/r5_gpu_suite.log
cp gpurun_out/parity_measured.json gpurun_out/r5_parity_measured.json 2>/dev/null
python bench.py 2>gpurun_out/r5_bench.err | grep '^{' > gpurun_out/r5_bench_try.json
python -c "
import json; d = json.load(open('gpurun_out/r5_bench_try.json'))
print('value', round(d['value'], 2), 'harness', d.get('harness_pairs_per_s'), 'batched', d.get('harness_batched_pairs_per_s'), (d.get('harness') or {}).get('batched'))"

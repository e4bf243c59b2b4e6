#!/bin/bash
# last pass of the round (GPU box, repo root): smoke, the whole GPU suite, the default bench line
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3
python -m pytest tests -q -m gpu > gpurun_out/r5_gpu_suite.log 2>&1; tail -n 4 gpurun_out/r5_gpu_suite.log
cp gpurun_out/parity_measured.json gpurun_out/r5_parity_measured.json
python bench.py 2>gpurun_out/r5_bench.err | grep '^{' > gpurun_out/r5_bench.json
python -c "
import json; d = json.load(open('gpurun_out/r5_bench.json'))
print('value', round(d['value'], 2), 'harness', d.get('harness_pairs_per_s'), 'batched', d.get('harness_batched_pairs_per_s'), 'frac', d['roofline']['frac'], d['roofline'].get('stale_profile_warning'), d['roofline']['traffic_source']['file'], d['roofline']['rocprof_source']['file'])"

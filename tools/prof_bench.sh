#!/bin/bash
# bench twice + rocprofv3 kernel summary of the same command (run on the GPU box from the repo root): bash tools/prof_bench.sh TAG [NLINES]
export TMPDIR=/tmp
T=${1:-x}; N=${2:-28}
for i in 1 2; do python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-corr-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('pairs/s', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 3), '1-in-flight', round(d.get('value_1_in_flight', 0), 2))"; done
rm -rf gpurun_out/prof_$T
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$T -o p -- python3 bench.py --no-cpu-baseline --no-corr-roofline --steps 40 > gpurun_out/prof_$T.log 2>&1
python tools/kernel_stats_summary.py $(find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -1) > gpurun_out/prof_${T}_summary.txt
head -$N gpurun_out/prof_${T}_summary.txt; tail -1 gpurun_out/prof_${T}_summary.txt

#!/bin/bash
# round 5: pairs/s over (batch, streams) + clean single-stream kernel table at batch 8 (GPU box, repo root)
export TMPDIR=/tmp
python -m pytest tests/test_harness_gpu.py -x -q -m gpu > gpurun_out/r5_harness_tests.log 2>&1
tail -n 3 gpurun_out/r5_harness_tests.log
for cfg in "1 3" "2 3" "2 2" "4 2" "4 3" "8 1" "8 2"; do
  set -- $cfg
  steps=$((160 / $1)); [ $steps -lt 20 ] && steps=20
  python bench.py --batch $1 --streams $2 --steps $steps --warmup 6 --no-cpu-baseline --no-corr-roofline --harness none 2>/dev/null | grep '^{' > gpurun_out/r5_sweep_b$1_s$2.json
  python -c "
import json; d = json.load(open('gpurun_out/r5_sweep_b$1_s$2.json')); print('batch $1 streams $2:', round(d['value'], 2), 'pairs/s', round(d['ms_per_step'], 3), 'ms/step; family frac', round(d['roofline']['frac'], 4))"
done
rm -rf gpurun_out/prof_b8s1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b8s1 -o p -- python3 bench.py --batch 8 --streams 1 --steps 20 --warmup 4 --no-cpu-baseline --no-corr-roofline --harness none > gpurun_out/prof_b8s1.log 2>&1
ST=$(find gpurun_out/prof_b8s1 -name "*kernel_stats.csv" | head -1)
cp $ST gpurun_out/r5_pre_bench_b8s1_kernel_stats.csv
python tools/kernel_stats_summary.py $ST > gpurun_out/r5_pre_b8s1_summary.txt
rm -rf gpurun_out/prof_b8s1
tail -n 1 gpurun_out/r5_pre_b8s1_summary.txt

#!/bin/bash
# round 5: kernel tables of the batch-1 and batch-8 bench commands (GPU box, repo root): bash tools/r5_prof_b8.sh
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py tests/test_tps_pipeline_gpu.py -x -q -m gpu -k "tps" -s > gpurun_out/r5_tps2.log 2>&1 || { tail -30 gpurun_out/r5_tps2.log; exit 1; }
grep "^\[tps" gpurun_out/r5_tps2.log
for B in 1 8; do
  S=$([ $B = 8 ] && echo "--batch 8 --streams 2 --steps 30 --warmup 4" || echo "--steps 80 --warmup 10")
  python bench.py $S --no-cpu-baseline --no-corr-roofline --harness none 2>/dev/null | grep '^{' > gpurun_out/r5_pre_bench_b$B.json
  python -c "
import json; d = json.load(open('gpurun_out/r5_pre_bench_b$B.json')); print('batch $B', round(d['value'], 2), 'pairs/s', round(d['ms_per_step'], 3), 'ms/step', d['roofline']['frac'])"
  rm -rf gpurun_out/prof_b$B
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$B -o p -- python3 bench.py $S --no-cpu-baseline --no-corr-roofline --harness none > gpurun_out/prof_b$B.log 2>&1
  ST=$(find gpurun_out/prof_b$B -name "*kernel_stats.csv" | head -1)
  cp $ST gpurun_out/r5_pre_bench_b${B}_kernel_stats.csv
  python tools/kernel_stats_summary.py $ST > gpurun_out/r5_pre_b${B}_summary.txt
  rm -rf gpurun_out/prof_b$B
done
tail -1 gpurun_out/r5_pre_b1_summary.txt gpurun_out/r5_pre_b8_summary.txt

#!/bin/bash
# rocprofv3 kernel summary of bench.py under two settings of one environment variable (GPU box, repo root): bash tools/prof_ab.sh VAR A B
export TMPDIR=/tmp
V=$1; shift
for val in "$@"; do
  rm -rf gpurun_out/prof_ab_$val
  export $V=$val
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_$val -o p -- python3 bench.py --no-cpu-baseline --no-corr-roofline --harness none --steps 40 > gpurun_out/prof_ab_$val.log 2>&1
  echo "== $V=$val"
  python tools/kernel_stats_summary.py $(find gpurun_out/prof_ab_$val -name "*kernel_stats.csv" | head -1) > gpurun_out/prof_ab_${V}_$val.txt
  head -14 gpurun_out/prof_ab_${V}_$val.txt; tail -1 gpurun_out/prof_ab_${V}_$val.txt
done

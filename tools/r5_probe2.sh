#!/bin/bash
export TMPDIR=/tmp
ST_PERSIST_SLOTS=512 python tools/persist_probe.py 2>&1 | grep ST_PERSIST
ST_PERSIST_SLOTS=256 python tools/persist_probe.py 2>&1 | grep ST_PERSIST
bash tools/dma_gemm_pmc.sh > gpurun_out/dma_pmc.log 2>&1; tail -n 30 gpurun_out/r5_dma_gemm_sq_counters.txt
rm -rf gpurun_out/prof_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_out -o p -- python3 tools/bench_out_harness.py 16 > gpurun_out/prof_out.log 2>&1
ST=$(find gpurun_out/prof_out -name "*kernel_stats.csv" | head -1)
cp $ST gpurun_out/r5_pre_out_harness_kernel_stats.csv
python tools/kernel_stats_summary.py $ST 128 > gpurun_out/r5_pre_out_harness_summary.txt
rm -rf gpurun_out/prof_out
tail -n 1 gpurun_out/r5_pre_out_harness_summary.txt

#!/bin/bash
# final measurement pass of the round (GPU box, repo root): bench lines + rocprof kernel stats + per-shape table with PMC traffic
export TMPDIR=/tmp
set -x
python bench.py --batch 8 --streams 2 --steps 30 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r3_bench_batch8.json
python bench.py --workload 1024 --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r3_bench_1024.json
rm -rf gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o p -- python3 bench.py --no-cpu-baseline --no-corr-roofline > gpurun_out/prof_final.log 2>&1
S=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1)
cp $S gpurun_out/r3_bench_kernel_stats.csv
python tools/kernel_stats_summary.py $S $(python - <<PY
import csv
rows=list(csv.DictReader(open("$S")))
print(next(int(r["Calls"]) for r in rows if "flow_warp_kernel" in r["Name"]))
PY
) gpurun_out/r3_kernel_summary.json > gpurun_out/r3_kernel_summary.txt
python tools/trace_overlap.py $(find gpurun_out/prof_final -name "*kernel_trace.csv" | head -1) > gpurun_out/r3_trace_overlap.txt
tail -3 gpurun_out/r3_kernel_summary.txt
bash tools/run_pmc_shapes.sh

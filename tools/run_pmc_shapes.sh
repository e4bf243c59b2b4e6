#!/bin/bash
# GEMM per-shape table with PMC traffic (run on the GPU box from the repo root): timing pass, two rocprofv3 --pmc passes
# (one counter each, --kernel-trace only), join.  Outputs under gpurun_out/; copy ${TAG}_gemm_shapes.csv / r3_traffic.json to profiles/.
set -e
TAG=${1:-r5}
export TMPDIR=/tmp
O=gpurun_out/shapes
mkdir -p $O
python3 tools/gemm_shapes_csv.py $O/${TAG}_gemm_shapes_timing.csv --launches $O/launches.json > $O/timing.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 tools/gemm_shapes_csv.py $O/unused.csv --profile-only > $O/pmc_$c.log 2>&1
done
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/gemm_shapes_csv.py $O/${TAG}_gemm_shapes.csv --join $O/launches.json "$F" "$W" > $O/join.log 2>&1
tail -5 $O/join.log

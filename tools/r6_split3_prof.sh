#!/bin/bash
# round-6 measurement pass of the split3 kernel (GPU box, repo root): bash tools/r6_split3_prof.sh [TAG=r6] -> gpurun_out/${TAG}_split3_*.{json,txt}
export TMPDIR=/tmp
TAG=${1:-r6}
# round 6: the split3 kernel -- stand-alone probe (gate shapes + the decoder's shapes), SQ counters, in-kernel clock / workgroup timeline, neighbour stress
python tools/split3_probe.py --iters 50 --tiles 32,34 --shapes conv1x5,aggregate,corr,zr5x1,q1x5,convc2,convf2,conv3x3,fh1 --json gpurun_out/${TAG}_split3_probe.json > gpurun_out/${TAG}_split3_probe.log 2>&1
grep "best split3" gpurun_out/${TAG}_split3_probe.log
bash tools/split3_pmc.sh > /dev/null 2>&1; cp gpurun_out/r6_split3_sq_counters.txt gpurun_out/${TAG}_split3_sq_counters.txt 2>/dev/null
ST_SPLIT3_DIAG=4 python tools/split3_clock.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${TAG}_split3_clock.txt
ST_SPLIT3_DIAG=2 python tools/split3_probe.py --iters 50 --tiles 34 --shapes conv1x5 2>/dev/null | grep "tile 34" | sed 's/^/MFMA-only (ST_SPLIT3_DIAG=2): /' >> gpurun_out/${TAG}_split3_clock.txt
ST_SPLIT3_DIAG=1 python tools/split3_probe.py --iters 50 --tiles 34 --shapes conv1x5 2>/dev/null | grep "tile 34" | sed 's/^/DMA-only (ST_SPLIT3_DIAG=1): /' >> gpurun_out/${TAG}_split3_clock.txt
cat gpurun_out/${TAG}_split3_clock.txt
python tools/neighbour_stress.py split3 100 2>/dev/null | tail -2 > gpurun_out/${TAG}_neighbour_stress.txt; cat gpurun_out/${TAG}_neighbour_stress.txt
for s in 0 1; do ST_SPLIT3=$s python tools/decoder_bench.py 2>/dev/null | tail -1; done > gpurun_out/${TAG}_decoder_bench.txt; cat gpurun_out/${TAG}_decoder_bench.txt
bash tools/r6_ab.sh "ST_SPLIT3=0" "ST_SPLIT3=1" > gpurun_out/${TAG}_split3_ab.txt 2>/dev/null; cat gpurun_out/${TAG}_split3_ab.txt

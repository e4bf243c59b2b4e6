#!/bin/bash
# MFMA utilisation of ONE eager forward (512x512 test_eval), per kernel and for the whole forward, from SQ counters (GPU box, repo root):
#   bash tools/mfma_util.sh > gpurun_out/r4_mfma_utilisation.txt
# Two rocprofv3 passes of the same command (counters in their own runs, --kernel-trace only): SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES +
# GRBM_GUI_ACTIVE, then SQ_INSTS_MFMA + SQ_INSTS_VALU + SQ_WAVES.  Kernel durations come from the first pass's kernel trace.
export TMPDIR=/tmp
O=gpurun_out/mfma_util
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o pmc -- python3 tools/gemm_shapes_csv.py $O/unused.csv --profile-only > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/p2 -o pmc -- python3 tools/gemm_shapes_csv.py $O/unused.csv --profile-only > $O/p2.log 2>&1
python3 tools/mfma_util.py $O

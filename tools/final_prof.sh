#!/bin/bash
# final measurement pass of the round (GPU box, repo root): bash tools/final_prof.sh [TAG=r5]
# bench lines (configs[1] with the product harness, configs[2], configs[3], 2-rank gloo rehearsal) + rocprof kernel stats of the same
# command + trace overlap + per-shape table with PMC traffic + SQ counters of the fused MLP kernel.  Everything lands under gpurun_out/;
# copy what is to be judged into profiles/.
export TMPDIR=/tmp
TAG=${1:-r6}
set -x
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | grep '^{' > gpurun_out/${TAG}_bench_nocpu.json
python bench.py --batch 8 --streams 2 --steps 30 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_batch8.json
# clean kernel table of the batch-8 forward (one stream: rocprof durations are then stand-alone; two streams overlap inside the trace)
rm -rf gpurun_out/prof_b8
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b8 -o p -- python3 bench.py --batch 8 --streams 1 --steps 20 --warmup 4 --no-cpu-baseline --no-corr-roofline --harness none > gpurun_out/prof_b8.log 2>&1
S8=$(find gpurun_out/prof_b8 -name "*kernel_stats.csv" | head -1)
cp $S8 gpurun_out/${TAG}_bench_batch8_kernel_stats.csv
python tools/kernel_stats_summary.py $S8 > gpurun_out/${TAG}_bench_batch8_kernel_summary.txt
rm -rf gpurun_out/prof_b8
# pairs/s over (batch, streams)
for cfg in "1 3" "2 3" "4 2" "4 3" "8 1" "8 2"; do set -- $cfg; steps=$((160 / $1)); [ $steps -lt 20 ] && steps=20
  python bench.py --batch $1 --streams $2 --steps $steps --warmup 6 --no-cpu-baseline --no-corr-roofline --harness none 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('batch $1 streams $2:', round(d['value'], 2), 'pairs/s', round(d['ms_per_step'], 3), 'ms/step; GEMM family frac', round(d['roofline']['frac'], 4))"
done > gpurun_out/${TAG}_batch_sweep.txt
cat gpurun_out/${TAG}_batch_sweep.txt
# the product harness with 2 pairs per forward (the sweep's best point) beside the default bench line's 4
python bench.py --no-cpu-baseline --no-corr-roofline --harness-batch 2 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_harness_batch2.json
python -c "
import json
for f in ('${TAG}_bench_nocpu', '${TAG}_bench_harness_batch2'):
    d = json.load(open('gpurun_out/' + f + '.json')); print(f, 'value', round(d['value'], 2), 'harness', round(d['harness_pairs_per_s'], 2), 'batched', round(d['harness_batched_pairs_per_s'], 2))
" > gpurun_out/${TAG}_harness_batch.txt; cat gpurun_out/${TAG}_harness_batch.txt
python bench.py --workload 1024 --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_1024.json
rm -rf gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o p -- python3 bench.py --no-cpu-baseline --no-corr-roofline --harness none > gpurun_out/prof_final.log 2>&1
S=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1)
cp $S gpurun_out/${TAG}_bench_kernel_stats.csv
python tools/kernel_stats_summary.py $S $(python - <<PY
import csv
rows=list(csv.DictReader(open("$S")))
print(next(int(r["Calls"]) for r in rows if "flow_warp_kernel" in r["Name"]))
PY
) gpurun_out/${TAG}_kernel_summary.json > gpurun_out/${TAG}_kernel_summary.txt
python tools/trace_overlap.py $(find gpurun_out/prof_final -name "*kernel_trace.csv" | head -1) > gpurun_out/${TAG}_trace_overlap.txt
tail -3 gpurun_out/${TAG}_kernel_summary.txt
bash tools/run_pmc_shapes.sh $TAG
bash tools/dma_gemm_pmc.sh > /dev/null 2>&1
tail -n 3 gpurun_out/r5_dma_gemm_sq_counters.txt
bash tools/mfma_util.sh > gpurun_out/${TAG}_mfma_utilisation.txt 2>/dev/null
bash tools/mlp_split3_pmc.sh > /dev/null 2>&1
python tools/mlp_split3_probe.py --json gpurun_out/${TAG}_mlp_split3_probe.json > gpurun_out/${TAG}_mlp_split3_probe.txt 2>/dev/null
ST_MLP3_DIAG=1 python tools/mlp_split3_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_mlp_split3_diag.txt
tools/probes/_bin/mbc > gpurun_out/${TAG}_mfma_bf16_chain.txt 2>/dev/null
tail -n 3 gpurun_out/${TAG}_mfma_utilisation.txt
python tools/bench_out_harness.py 48 2>/dev/null | grep '^{' > gpurun_out/${TAG}_out_harness.json; cat gpurun_out/${TAG}_out_harness.json
# same-box A/B of the round's switches (each configuration twice)
bash tools/r6_ab.sh "ST_SPLIT3=0" "ST_SPLIT3=1 ST_S3_MLP=0 ST_SPLIT3_KPAR=0 ST_S3_LIN=0 ST_S3_CHAIN=0 ST_S3_PE_TAIL=0" "ST_SPLIT3=1 ST_SPLIT3_KPAR=0 ST_S3_LIN=0 ST_S3_CHAIN=0 ST_S3_PE_TAIL=0" "ST_SPLIT3=1 ST_S3_LIN=0 ST_S3_CHAIN=0 ST_S3_PE_TAIL=0" "ST_SPLIT3=1 ST_S3_PE_TAIL=0" "ST_SPLIT3=1" > gpurun_out/${TAG}_ab_switches.txt 2>/dev/null; cat gpurun_out/${TAG}_ab_switches.txt
python bench.py 2>>gpurun_out/${TAG}_bench.err | grep '^{' > gpurun_out/${TAG}_bench.json
# 2-rank rehearsal of the launcher / sharding / all-gather on the 1-GPU box (both ranks on cuda:0, gloo): bounded, last
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --steps 40 --warmup 6 --no-cpu-baseline --no-corr-roofline --harness-pairs 48 2>gpurun_out/${TAG}_bench_2rank.err | grep '^{' > gpurun_out/${TAG}_bench_2rank_gloo_share_gpu.json
python -c "
import json
for f in ('${TAG}_bench', '${TAG}_bench_batch8', '${TAG}_bench_1024', '${TAG}_bench_2rank_gloo_share_gpu'):
    try: d = json.load(open('gpurun_out/' + f + '.json'))
    except Exception as e: print(f, 'missing', e); continue
    print(f, round(d['value'], 2), d.get('harness_pairs_per_s'), d.get('value_1_in_flight'), round(d['roofline']['frac'], 4))
"

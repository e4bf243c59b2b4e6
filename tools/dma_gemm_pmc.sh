#!/bin/bash
# SQ counters of conv_gemm_dma_kernel<2,2,1,1,4> on 8192x256x1920 (1x5 conv) and 8192x256x1152 (3x3 conv): separate rocprofv3 --pmc passes
# (GPU box, repo root) -> gpurun_out/r5_dma_gemm_sq_counters.txt
export TMPDIR=/tmp
O=gpurun_out/dma_pmc
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o pmc -- python3 tools/dma_gemm_pmc.py > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $O/p$i.log)"
done
python3 - <<'PY' | tee gpurun_out/r5_dma_gemm_sq_counters.txt
import csv, glob, collections
print("# SQ counters of conv_gemm_dma_kernel<2,2,1,1,4,false,false> (the dominant kernel: 197 launches, 6.3 ms of a pair's 14.06 ms) on the two shapes VERDICT r4 item 4 names")
print("# rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 tools/dma_gemm_pmc.py (tools/dma_gemm_pmc.sh); last of 5 launches per shape")
print("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES are cycles summed over the 1024 SIMDs (64 per v_mfma_f32_32x32x2_f32)")
per = collections.OrderedDict()          # shape index (0: first 5 dispatches of the kernel, 1: next 5) -> counters
durs = {}
for p in (1, 2, 3, 4):
    for f in glob.glob(f"gpurun_out/dma_pmc/p{p}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "conv_gemm_dma_kernel" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        for r in rows:
            k = ids.index(int(r["Dispatch_Id"]))
            if k % 5 == 4: per.setdefault(k // 5, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if p == 1:
        for f in glob.glob(f"gpurun_out/dma_pmc/p{p}/**/*kernel_trace.csv", recursive=True):
            rows = sorted((r for r in csv.DictReader(open(f)) if "conv_gemm_dma_kernel" in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))
            for k, r in enumerate(rows):
                if k % 5 == 4: durs[k // 5] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
names = {0: "M=8192 N=256 K=1920 (1x5 conv, Cin 384: SepConvGRU z|r)", 1: "M=8192 N=256 K=1152 (3x3 conv, Cin 128)"}
flops = {0: 2 * 8192 * 256 * 1920, 1: 2 * 8192 * 256 * 1152}
for k, agg in per.items():
    dur = durs.get(k)
    wc = agg.get("SQ_WAVE_CYCLES", 1)
    print(f"\n{names.get(k, k)}: {dur} us under the profiler ({flops[k] / dur / 1e6:.1f} TFLOP/s = {flops[k] / dur / 1e6 / 157.3:.3f} of the 157.3 TFLOP/s nominal peak)")
    for n, v in agg.items():
        extra = f"  ({v / wc:.3f} of SQ_WAVE_CYCLES)" if n.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES", "SQ_INST_LEVEL")) else ""
        print(f"   {n:28s} {v:14.0f}{extra}")
    if dur and "SQ_VALU_MFMA_BUSY_CYCLES" in agg:
        clk = agg.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur * 1e-6) / 1e9
        busy = agg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (dur * 1e-6) / 1e9
        print(f"   -> MFMA pipe busy {busy:.3f} GHz-equivalents per SIMD; shader clock over the dispatch {clk:.3f} GHz; busy / clock = {busy / clk if clk else float('nan'):.3f}; busy / 2.4 GHz = {busy / 2.4:.3f}")
        nm = agg.get("SQ_INSTS_MFMA", 0)
        if nm:
            print(f"   -> per 16 MFMAs (one 32-deep K step of a wave): VALU {16 * (agg.get('SQ_INSTS_VALU', 0) - nm) / nm:.2f} (MFMAs excluded), SALU {16 * agg.get('SQ_INSTS_SALU', 0) / nm:.2f}, "
                  f"LDS {16 * agg.get('SQ_INSTS_LDS', 0) / nm:.2f}, VMEM {16 * agg.get('SQ_INSTS_VMEM', 0) / nm:.2f}")
PY

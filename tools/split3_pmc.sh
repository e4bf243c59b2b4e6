#!/bin/bash
# SQ counters of conv_gemm_split3_kernel on 8192x256x1920 (1x5 conv), tiles 64x64 and 128x64: separate rocprofv3 --pmc passes
# (GPU box, repo root) -> gpurun_out/r6_split3_sq_counters.txt.  ST_SPLIT3_DIAG=1|2 in the environment profiles the ingest-only / MFMA-only variants.
export TMPDIR=/tmp
O=gpurun_out/s3_pmc${ST_SPLIT3_DIAG}
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o pmc -- python3 tools/split3_pmc.py > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $O/p$i.log)"
done
python3 - $O <<'PY' | tee gpurun_out/r6_split3_sq_counters${ST_SPLIT3_DIAG}.txt
import csv, glob, collections, sys, os
O = sys.argv[1]
print(f"# SQ counters of conv_gemm_split3_kernel on M=8192 N=256 K=1920 (1x5 conv, Cin 384), ST_SPLIT3_DIAG={os.environ.get('ST_SPLIT3_DIAG', '0')}")
print("# rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 tools/split3_pmc.py (tools/split3_pmc.sh); last of 5 launches per tile config")
per, durs = collections.OrderedDict(), {}
for p in (1, 2, 3):
    for f in glob.glob(f"{O}/p{p}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "conv_gemm_split3_kernel" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        for r in rows:
            k = ids.index(int(r["Dispatch_Id"]))
            if k % 5 == 4: per.setdefault(k // 5, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if p == 1:
        for f in glob.glob(f"{O}/p{p}/**/*kernel_trace.csv", recursive=True):
            rows = sorted((r for r in csv.DictReader(open(f)) if "conv_gemm_split3_kernel" in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))
            for k, r in enumerate(rows):
                if k % 5 == 4: durs[k // 5] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
names = {0: "tile 64x64, 3-stage ring, two workgroups per CU", 1: "tile 128x64, 4-stage ring, one workgroup per CU"}
flop = 2 * 8192 * 256 * 1920
for k, agg in per.items():
    dur = durs.get(k)
    wc = agg.get("SQ_WAVE_CYCLES", 1)
    print(f"\n{names.get(k, k)}: {dur} us under the profiler ({flop / dur / 1e6:.1f} fp32-equivalent TFLOP/s)")
    for n, v in agg.items():
        extra = f"  ({v / wc:.3f} of SQ_WAVE_CYCLES)" if n.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES", "SQ_INST_LEVEL")) else ""
        print(f"   {n:28s} {v:14.0f}{extra}")
    if dur and "SQ_VALU_MFMA_BUSY_CYCLES" in agg:
        clk = agg.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur * 1e-6) / 1e9
        busy = agg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (dur * 1e-6) / 1e9
        print(f"   -> MFMA pipe busy {busy:.3f} GHz-equivalents per SIMD; shader clock over the dispatch {clk:.3f} GHz; busy / clock = {busy / clk if clk else float('nan'):.3f}")
        nm = agg.get("SQ_INSTS_MFMA", 0)
        if nm:
            print(f"   -> per 12 MFMAs: VALU {12 * (agg.get('SQ_INSTS_VALU', 0) - nm) / nm:.2f} (MFMAs excluded), SALU {12 * agg.get('SQ_INSTS_SALU', 0) / nm:.2f}, "
                  f"LDS {12 * agg.get('SQ_INSTS_LDS', 0) / nm:.2f}, VMEM {12 * agg.get('SQ_INSTS_VMEM', 0) / nm:.2f}; ideal MFMA time {nm / 1024 * 32 / 4 / 1e3:.1f} kcycles per SIMD")
PY

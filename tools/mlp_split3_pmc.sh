#!/bin/bash
# SQ counters of rowmlp128_kernel<true> and rowmlp128_split3_kernel<true> on M = 65536, hidden 512: separate rocprofv3 --pmc passes
# (GPU box, repo root) -> gpurun_out/r6_mlp_split3_sq_counters.txt
export TMPDIR=/tmp
O=gpurun_out/mlp3_pmc
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o pmc -- python3 tools/mlp_split3_pmc.py > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $O/p$i.log)"
done
python3 - $O <<'PY' | tee gpurun_out/r6_mlp_split3_sq_counters.txt
import csv, glob, collections, sys, os
O = sys.argv[1]
print("# SQ counters of the C = 128 block tail (projection + LayerNorm + fc1 + GELU + fc2 + residuals), M = 65536, hidden 512: fp32-MFMA kernel vs split3 kernel")
print("# rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 tools/mlp_split3_pmc.py (tools/mlp_split3_pmc.sh); last of 5 launches each")
for kern, mf in (("rowmlp128_kernel", 64), ("rowmlp128_split3_kernel", 32)):
    agg, dur = collections.OrderedDict(), None
    for p in (1, 2, 3):
        for f in glob.glob(f"{O}/p{p}/**/*counter_collection.csv", recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void " + kern + "<") or r["Kernel_Name"].startswith(kern + "<")]
            ids = sorted({int(r["Dispatch_Id"]) for r in rows})
            for r in rows:
                if ids.index(int(r["Dispatch_Id"])) == 4: agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if p == 1:
            for f in glob.glob(f"{O}/p{p}/**/*kernel_trace.csv", recursive=True):
                rows = sorted((r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void " + kern + "<") or r["Kernel_Name"].startswith(kern + "<")), key=lambda r: int(r["Dispatch_Id"]))
                if len(rows) >= 5: dur = (int(rows[4]["End_Timestamp"]) - int(rows[4]["Start_Timestamp"])) / 1e3
    if not agg:
        print(kern, "no rows"); continue
    wc = agg.get("SQ_WAVE_CYCLES", 1)
    flop = 2 * 65536 * 128 * (1024 + 128)
    print(f"\n{kern}: {dur} us under the profiler ({flop / dur / 1e6:.1f} fp32-equivalent TFLOP/s)")
    for n, v in agg.items():
        extra = f"  ({v / wc:.3f} of SQ_WAVE_CYCLES)" if n.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")) else ""
        print(f"   {n:28s} {v:14.0f}{extra}")
    clk = agg.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur * 1e-6) / 1e9
    busy = agg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (dur * 1e-6) / 1e9
    nm = agg.get("SQ_INSTS_MFMA", 0)
    print(f"   -> MFMA pipe busy {busy:.3f} GHz-equivalents per SIMD; clock from GRBM_GUI_ACTIVE {clk:.3f} GHz; busy / clock = {busy / clk if clk else float('nan'):.3f}")
    if nm:
        print(f"   -> per 96 MFMAs: VALU {96 * (agg.get('SQ_INSTS_VALU', 0) - nm) / nm:.1f} (MFMAs excluded), SALU {96 * agg.get('SQ_INSTS_SALU', 0) / nm:.1f}, LDS {96 * agg.get('SQ_INSTS_LDS', 0) / nm:.1f}, "
              f"VMEM {96 * agg.get('SQ_INSTS_VMEM', 0) / nm:.1f}; MFMA instructions x {mf} cycles = {nm / 1024 * mf / 1e3:.1f} kcycles per SIMD")
PY

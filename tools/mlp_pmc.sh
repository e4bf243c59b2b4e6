#!/bin/bash
# SQ counters of rowmlp128_kernel (st_mlp128, M = 65536, hidden 512): three rocprofv3 --pmc passes (GPU box, repo root) -> gpurun_out/r4_rowmlp_sq_counters.txt
export TMPDIR=/tmp
O=gpurun_out/mlp_pmc
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o pmc -- python3 tools/mlp_pmc.py > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY' | tee gpurun_out/r4_rowmlp_sq_counters.txt
import csv, glob
agg, dur = {}, None
for p in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/mlp_pmc/p{p}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "rowmlp128" in r["Kernel_Name"]]
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last: agg[r["Counter_Name"]] = float(r["Counter_Value"])
    for f in glob.glob(f"gpurun_out/mlp_pmc/p{p}/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "rowmlp128" in r["Kernel_Name"]]
        if rows and p == 1:
            r = max(rows, key=lambda r: int(r["Dispatch_Id"])); dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
wc = agg.get("SQ_WAVE_CYCLES", 1)
print(f"rowmlp128_kernel (st_mlp128: LN -> fc1 128->512 + GELU -> fc2 512->128 + residual, M=65536): {dur} us under the profiler (rocprofv3 --pmc, 3 passes, last of 6 launches)")
for n, v in agg.items():
    extra = f"  ({v / wc:.3f} of SQ_WAVE_CYCLES)" if n.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")) else ""
    print(f"   {n:28s} {v:14.0f}{extra}")
if dur:
    print(f"   MFMA busy = {agg['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (dur * 1e-6) / 1e9:.2f} GHz-equivalents per SIMD; clock {agg.get('GRBM_GUI_ACTIVE', 0) / 8 / (dur * 1e-6) / 1e9:.2f} GHz")
    nm = agg.get("SQ_INSTS_MFMA", 0)
    if nm: print(f"   VALU instructions per 64 MFMAs: {64 * agg.get('SQ_INSTS_VALU', 0) / nm:.0f} (the count includes the MFMAs themselves)")
PY

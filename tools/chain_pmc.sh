#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/chain_pmc
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o pmc -- python3 tools/chain_pmc.py > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob
agg, dur = {}, None
for p in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/chain_pmc/p{p}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "rowchain" in r["Kernel_Name"]]
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last: agg[r["Counter_Name"]] = float(r["Counter_Value"])
    for f in glob.glob(f"gpurun_out/chain_pmc/p{p}/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "rowchain" in r["Kernel_Name"]]
        if rows and p == 1:
            r = max(rows, key=lambda r: int(r["Dispatch_Id"])); dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
wc = agg.get("SQ_WAVE_CYCLES", 1)
print(f"rowchain128 (3 layers, M=65536): {dur} us under the profiler")
for n, v in agg.items():
    extra = f"  ({v / wc:.3f} of SQ_WAVE_CYCLES)" if n.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")) else ""
    print(f"   {n:28s} {v:14.0f}{extra}")
if dur: print(f"   MFMA busy = {agg['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (dur * 1e-6) / 1e9:.2f} GHz-equivalents per SIMD; clock {agg.get('GRBM_GUI_ACTIVE', 0) / 8 / (dur * 1e-6) / 1e9:.2f} GHz")
PY

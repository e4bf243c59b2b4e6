#!/bin/bash
# SQ counters of the row-streaming kernel (tile 20) and the LDS-DMA kernel (tile 13): bash tools/rs_pmc.sh  -> gpurun_out/rs_pmc/
export TMPDIR=/tmp
O=gpurun_out/rs_pmc
mkdir -p $O
for t in 20 13; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA" \
             "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/t${t}_p$i -o pmc -- python3 tools/rs_pmc.py $t > $O/t${t}_p$i.log 2>&1 || echo "pass $t $i failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, re
for t in (20, 13):
    agg = collections.OrderedDict()
    dur = {}
    for p in (1, 2, 3):
        for f in glob.glob(f"gpurun_out/rs_pmc/t{t}_p{p}/**/*counter_collection.csv", recursive=True):
            rows = list(csv.DictReader(open(f)))
            gem = [r for r in rows if "gemm" in r["Kernel_Name"]]
            ids = sorted({int(r["Dispatch_Id"]) for r in gem})
            # 3 shapes x 4 launches: keep the last launch of each shape
            keep = {ids[3]: 0, ids[7]: 1, ids[11]: 2} if len(ids) >= 12 else {}
            for r in gem:
                k = keep.get(int(r["Dispatch_Id"]))
                if k is not None:
                    agg.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for f in glob.glob(f"gpurun_out/rs_pmc/t{t}_p{p}/**/*kernel_trace.csv", recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            if len(rows) >= 12 and p == 1:
                for k, i in ((0, 3), (1, 7), (2, 11)):
                    dur[k] = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    names = ["524288x128x128", "65536x128x128", "65536x512x128"]
    for k, c in agg.items():
        wc = c.get("SQ_WAVE_CYCLES", 1)
        print(f"tile {t} {names[k]}: {dur.get(k, 0):.1f} us under the profiler")
        for n, v in c.items():
            extra = f"  ({v / wc:.3f} of SQ_WAVE_CYCLES)" if n.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")) else ""
            print(f"   {n:28s} {v:14.0f}{extra}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and dur.get(k):
            print(f"   MFMA busy = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (dur[k] * 1e-6) / 1e9:.2f} GHz-equivalents per SIMD")
PY

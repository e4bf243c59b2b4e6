"""Aggregate tools/mfma_util.sh's counter passes: MFMA utilisation per kernel and for the whole forward.

SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip's 1024 SIMDs, the cycles in which a SIMD's matrix pipe is busy (64 per
v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32: MI355X_MICROARCH.md cycle constants), so
    utilisation = MFMA_BUSY / (1024 x kernel duration x clock),   clock = GRBM_GUI_ACTIVE / 8 / duration  (sum over the 8 XCDs)
i.e. MFMA_BUSY / (128 x GRBM_GUI_ACTIVE).  The profiled command runs 2 warm-up forwards + 1: the last third of the dispatches is kept."""
import collections, csv, glob, re, sys
O = sys.argv[1]
def load(p):
    f = glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True)[0]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = rows.setdefault(int(r["Dispatch_Id"]), dict(name=re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()))
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    return rows
def durations(p):
    f = glob.glob(f"{O}/{p}/**/*kernel_trace.csv", recursive=True)[0]
    return {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f))}
a, b, dur = load("p1"), load("p2"), durations("p1")
ids = [i for i in a if "at::" not in a[i]["name"] and "spin" not in a[i]["name"]]
def last_forward(rows, ids):
    """a forward starts with the homography net's input prep: prep_image_kernel runs 4 times per forward, the first of them first"""
    pi = [k for k, i in enumerate(ids) if "prep_image" in rows[i]["name"]]
    return ids[pi[-4]:]
ids = last_forward(a, ids)                        # the last of the three forwards
agg = collections.OrderedDict()
for i in ids:
    r = a[i]
    g = agg.setdefault(r["name"][:64], dict(n=0, us=0.0, mfma=0.0, gui=0.0, insts=0.0))
    g["n"] += 1; g["us"] += dur.get(i, 0.0); g["mfma"] += r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); g["gui"] += r.get("GRBM_GUI_ACTIVE", 0.0)
bi = [i for i in b if "at::" not in b[i]["name"] and "spin" not in b[i]["name"]]
for i in last_forward(b, bi):
    g = agg.get(b[i]["name"][:64])
    if g: g["insts"] += b[i].get("SQ_INSTS_MFMA", 0.0)
T = sum(g["us"] for g in agg.values()); M = sum(g["mfma"] for g in agg.values()); G = sum(g["gui"] for g in agg.values())
print(f"# one eager forward(type=test_eval), 512x512, batch 1: {len(ids)} dispatches, {T / 1e3:.2f} ms of kernel time under the profiler")
print(f"# whole forward: the matrix pipes are busy {M / 1024 / (T * 1e-6) / 2.4e9:.3f} of the cycles a 2.4 GHz clock offers over the forward's kernel time, i.e. that fraction of")
print(f"#   the 157.3 TFLOP/s fp32-MFMA peak is occupied by matrix instructions (MFMA_BUSY / 1024 SIMDs / duration / 2.4 GHz; GRBM_GUI_ACTIVE / 8 / duration reads")
print(f"#   above the real clock on dispatches this short -- MI355X_MICROARCH.md -- so the per-kernel column is normalised to 2.4 GHz too)")
print(f"# {'kernel':64s} {'launches':>8s} {'ms':>7s} {'share':>6s} {'MFMA busy':>9s}")
for k, g in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
    if g["us"] / T < 0.002: continue
    print(f"  {k:64s} {g['n']:8d} {g['us'] / 1e3:7.3f} {g['us'] / T:6.3f} {g['mfma'] / 1024 / (g['us'] * 1e-6) / 2.4e9 if g['us'] else 0:9.3f}")

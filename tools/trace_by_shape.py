"""Per (kernel, grid) durations of a rocprofv3 kernel trace: python tools/trace_by_shape.py TRACE.csv [min_calls]"""
import csv, collections, re, sys
rows = csv.DictReader(open(sys.argv[1]))
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 20
agg = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:60]
    agg[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= minc:
        v = sorted(v)
        print(f"{sum(v) / tot * 100:5.1f} %  x{len(v):5d}  median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f}  {k[0]} grid ({k[1]},{k[2]},{k[3]})")

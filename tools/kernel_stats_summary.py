"""rocprofv3 --kernel-trace --stats summary -> ms per forward by kernel: python tools/kernel_stats_summary.py STATS.csv [FORWARDS]
FORWARDS defaults to the call count of flow_warp_kernel (one per forward)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:78]
fw = int(sys.argv[2]) if len(sys.argv) > 2 else next(int(r["Calls"]) for r in rows if "flow_warp_kernel" in r["Name"])
tot = 0.0
gemm = 0.0
print(f"forwards: {fw}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if "spin_kernel" in r["Name"]:      # torch.cuda._sleep of bench.py's instrumented step: not part of a forward
        continue
    ms = float(r["TotalDurationNs"]) / fw / 1e6
    tot += ms
    if any(k in r["Name"] for k in ("conv_gemm", "rowstream_gemm", "rowchain128", "rowmlp128", "rowlin128", "pe_tail_split3", "splitk_reduce", "narrow_conv", "skinny_gemm", "patch_c0c2")):
        gemm += ms
    if ms >= 0.004:
        print(f"{ms:8.3f} ms  x{int(r['Calls']) / fw:7.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {short(r['Name'])}")
print(f"total kernel time per forward {tot:.3f} ms; GEMM family {gemm:.3f} ms; other {tot - gemm:.3f} ms")
if len(sys.argv) > 3:          # python tools/kernel_stats_summary.py STATS.csv FORWARDS OUT.json
    import json
    json.dump(dict(forwards=fw, total_kernel_ms_per_forward=tot, gemm_family_ms_per_forward=gemm, other_kernels_ms_per_forward=tot - gemm,
                   source="rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-corr-roofline (kernel tracing serialises "
                          "the dispatches of the three streams -- tools/trace_overlap.py on the same trace, profiles/r4_trace_overlap.txt: one kernel "
                          "resident 93 % of the time, two 1 % -- so these are STAND-ALONE kernel durations; the timed region runs them overlapped)"), open(sys.argv[3], "w"), indent=1)

"""Per-shape roofline table of the fp32-MFMA GEMM family: one row per distinct st_conv_gemm launch shape of ONE step
(FlowHomoAdpater.forward(type="test_eval"), 512x512, batch 1) -> profiles/r2_gemm_shapes.csv.

    python tools/gemm_shapes_csv.py OUT.csv [--launches LAUNCHES.json]           # timing pass (HIP events, observer hook)
    python tools/gemm_shapes_csv.py OUT.csv --join LAUNCHES.json FETCH.csv WRITE.csv   # add PMC bytes per shape (no GPU)

Timing pass: two warm-up steps, then one eager step (enqueued behind a spin kernel; the median reading of an empty event bracket is
subtracted) with a HIP-event pair around every launch (library observer,
`st_gemm_last_plan` tells which kernel / tile / split-K / persistent walk the library chose).  With --launches the ordered
launch list is saved so that a later `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` run OF THE SAME
COMMAND (one counter per pass, MI355X_MICROARCH.md) can be joined: GEMM-family dispatches appear in the same order in
rocprofv3's counter_collection.csv (split-K launches are followed by their reducer, which is counted with them).
Columns: algorithmic bytes = A (input rows x Cin, each element once) + W + C, fp32; PMC bytes with the gfx950 corrections
(FETCH_SIZE KiB x 2, WRITE_SIZE KiB x 1)."""
import collections
import csv
import ctypes as C
import json
import re
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])

FAMILY = {0: "skinny_gemm_kernel", 1: "narrow_conv_kernel", 2: "conv_gemm_kernel", 3: "conv_gemm_dma_kernel", 4: "rowstream_gemm_kernel", 5: "rowchain128_kernel",
          6: "rowmlp128_kernel", 7: "patch_c0c2_kernel", 8: "conv_gemm_split3_kernel", 9: "rowmlp128_split3_kernel"}
PEAK = 157.3


def shape_key(r):
    return (r["M"], r["N"], r["K"], r["conv"], r["batch"], r["kernel"], r["tile"], r["split_k"], r["persist"], r["epi"])


def timing_pass(out_csv, launches_json=None, profile_only=False):
    import torch
    import stitch_amd
    from stitch_amd.data import structured_pair
    lib, GemmDesc = stitch_amd._lib.lib, stitch_amd._lib.GemmDesc
    cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    torch.manual_seed(1234)
    model = stitch_amd.build_model(cfg).cuda().eval()
    a, b = (t.cuda() for t in structured_pair(512, 512, seed=7))
    for _ in range(2):
        model(a, b, type="test_eval")
    torch.cuda.synchronize()
    rec, open_ev = [], []
    plan = (C.c_int32 * 4)()

    @C.CFUNCTYPE(None, C.POINTER(GemmDesc), C.c_void_p, C.c_int32, C.c_void_p)
    def observer(desc, stream, phase, user):
        ev = None
        if not profile_only:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream())
        if phase == 0:
            open_ev.append(ev)
            return
        d = desc.contents
        lib.st_gemm_last_plan(plan)
        nb = max(1, d.batch)
        conv = d.kh * d.kw > 1 or d.sh > 1
        a_rows = (d.M // max(1, d.Ho * d.Wo)) * d.H * d.W if conv else d.M
        w_bytes = 4.0 * nb * d.N * d.K
        if plan[0] in (5, 6):
            # fused row kernels (reported as M x 128 L x 128 / M x 2 hidden x 128): the intermediate activations never move --
            # A rows in, weights, 128-wide rows out (+ the residual rows re-read by the MLP kernel)
            alg = 4.0 * (d.M * 128 + d.M * 128) + w_bytes + (4.0 * d.M * 128 if plan[0] == 6 else 0.0)
        elif plan[0] == 9:
            # the block tail on the split3 kernel: fp32 rows in (+ the projection's residual rows) and out, weights as three bf16 planes
            w_bytes = 6.0 * d.N * d.K
            alg = 4.0 * 3 * d.M * 128 + w_bytes
        elif d.split3:
            # exact-split operands: three bf16 planes = 6 bytes per element of A and W
            w_bytes = 6.0 * nb * d.N * d.K
            alg = nb * (6.0 * a_rows * d.Cin + (0.0 if d.c_no_f32 else 4.0 * d.M * d.N)) + w_bytes
        else:
            alg = 4.0 * nb * (a_rows * d.Cin + d.M * d.N) + w_bytes
            if d.c_t:
                alg += 4.0 * nb * d.M * d.N                       # the transposed second store of st_corr_volume_both
        if d.c_planes:                                            # the result also leaves as planes (6 bytes per element)
            alg += 6.0 * nb * d.M * (d.N // 2 if d.epi == 5 else d.N)
        rec.append(dict(M=d.M, N=d.N, K=d.K, conv=f"{d.kh}x{d.kw}s{d.sh}" + (f"d{d.dh}" if d.dh > 1 else ""), batch=nb,
                        kernel=FAMILY.get(plan[0], "?"), tile=plan[1], split_k=plan[2], persist=plan[3], epi=d.epi,
                        flops=2.0 * d.M * d.N * d.K * nb, alg_bytes=alg, w_bytes=w_bytes,
                        ev=(open_ev.pop(), ev)))

    lib.st_set_gemm_observer(C.cast(observer, C.c_void_p), None)
    empties = []
    try:
        if not profile_only:
            # hold the stream back (~0.1 s spin kernel) while the host enqueues the eager step, as bench.py's instrumented step does: a
            # bracket then reads its kernel(s) + the event overhead, not the Python launch latency an idle GPU would wait for (without
            # it the first launches of the step read ~100 us whatever they do)
            torch.cuda._sleep(int(2.4e8))
        model(a, b, type="test_eval")
        if not profile_only:
            st = torch.cuda.current_stream()
            for _ in range(64):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); e1.record(st)
                empties.append((e0, e1))
        torch.cuda.synchronize()
    finally:
        lib.st_set_gemm_observer(None, None)
    ov = sorted(e0.elapsed_time(e1) for e0, e1 in empties)[len(empties) // 2] if empties else 0.0     # an EMPTY bracket's reading (ms)
    for r in rec:
        e0, e1 = r.pop("ev")
        r["us"] = 1e3 * max(0.0, e0.elapsed_time(e1) - ov) if e0 is not None else 0.0
    if launches_json:
        json.dump(rec, open(launches_json, "w"))
    if not profile_only:
        write_csv(out_csv, rec)


def merge_pairs(rec):
    """st_conv_gemm_pair: the observer reports the first member with an empty bracket (persist = 2) and the second (persist = 3) with the
    pair's single dispatch and all of its time: one row for the launch, N = 'N0+N1', K of the first member, FLOPs and bytes summed."""
    out, i = [], 0
    while i < len(rec):
        r = rec[i]
        if r["persist"] == 2 and i + 1 < len(rec) and rec[i + 1]["persist"] == 3:
            m = dict(r)
            o = rec[i + 1]
            m.update(N=f"{r['N']}+{o['N']}", kernel="conv_gemm_split3_pair_kernel" if "split3" in r["kernel"] else "conv_gemm_dma_pair_kernel", persist=0, us=r["us"] + o["us"], flops=r["flops"] + o["flops"],
                     alg_bytes=r["alg_bytes"] + o["alg_bytes"], w_bytes=r.get("w_bytes", 0.0) + o.get("w_bytes", 0.0))
            if "fetch_bytes" in o:
                m["fetch_bytes"] = r.get("fetch_bytes", 0.0) + o["fetch_bytes"]
                m["write_bytes"] = r.get("write_bytes", 0.0) + o["write_bytes"]
            out.append(m)
            i += 2
        else:
            out.append(r)
            i += 1
    return out


def write_csv(out_csv, rec):
    agg = collections.OrderedDict()
    for r in merge_pairs(rec):
        g = agg.setdefault(shape_key(r), dict(launches=0, us=0.0, flops=0.0, alg=0.0, fetch=0.0, write=0.0, pmc=0, w=0.0))
        g["w"] += r.get("w_bytes", 0.0)
        g["launches"] += 1
        g["us"] += r["us"]
        g["flops"] += r["flops"]
        g["alg"] += r["alg_bytes"]
        if "fetch_bytes" in r:
            g["fetch"] += r["fetch_bytes"]
            g["write"] += r["write_bytes"]
            g["pmc"] += 1
    tot_us = sum(g["us"] for g in agg.values())
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["M", "N", "K", "conv", "batch", "kernel", "tile_cfg", "split_k", "persistent", "epilogue", "launches_per_step",
                    "us_per_launch", "ms_per_step", "share_of_gemm_time", "TFLOP/s", "frac_of_fp32_mfma_peak", "GFLOP_per_launch",
                    "algorithmic_MB_per_launch", "pmc_fetch_MB_per_launch", "pmc_write_MB_per_launch", "pmc_over_algorithmic",
                    "HBM_GB/s_algorithmic", "weights_x7_MB_per_launch", "pmc_over_algorithmic_8xcd"])
        for key, g in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
            n = g["launches"]
            tf = g["flops"] / g["us"] / 1e6 if g["us"] else 0.0
            pmc = (g["fetch"] + g["write"]) if g["pmc"] else None
            w.writerow(list(key[:10]) + [n, f"{g['us'] / n:.1f}", f"{g['us'] / 1e3:.3f}", f"{g['us'] / tot_us:.4f}" if tot_us else "",
                                        f"{tf:.1f}", f"{tf / PEAK:.3f}", f"{g['flops'] / n / 1e9:.3f}", f"{g['alg'] / n / 1e6:.2f}",
                                        "" if pmc is None else f"{g['fetch'] / n / 1e6:.2f}", "" if pmc is None else f"{g['write'] / n / 1e6:.2f}",
                                        "" if pmc is None else f"{pmc / g['alg']:.2f}", f"{g['alg'] / g['us'] / 1e3:.0f}" if g["us"] else "",
                                        # the 8 XCDs have private L2s: every XCD that runs tiles of a launch fetches its own copy of the weight panel, so
                                        # the floor of the L2-miss traffic (what FETCH_SIZE counts) is A + C + 8 W, not A + C + W
                                        f"{7 * g['w'] / n / 1e6:.2f}", "" if pmc is None else f"{pmc / (g['alg'] + 7 * g['w']):.2f}"])
        tf = sum(g["flops"] for g in agg.values()) / tot_us / 1e6 if tot_us else 0.0
        w.writerow(["TOTAL", "", "", "", "", "", "", "", "", "", sum(g["launches"] for g in agg.values()), "", f"{tot_us / 1e3:.3f}", "1.0",
                    f"{tf:.1f}", f"{tf / PEAK:.3f}", "", f"{sum(g['alg'] for g in agg.values()) / 1e6:.1f}",
                    f"{sum(g['fetch'] for g in agg.values()) / 1e6:.1f}", f"{sum(g['write'] for g in agg.values()) / 1e6:.1f}", "", "",
                    f"{7 * sum(g['w'] for g in agg.values()) / 1e6:.1f}", ""])
    print(open(out_csv).read())


def join(out_csv, launches_json, fetch_csv, write_csv_path):
    rec = json.load(open(launches_json))

    def gemm_rows(path):
        rows = []
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
            if any(s in k for s in ("conv_gemm", "skinny_gemm", "narrow_conv", "splitk_reduce", "rowstream_gemm", "rowchain128", "rowmlp128", "patch_c0c2")):
                rows.append((int(r.get("Dispatch_Id", len(rows))), k, float(r["Counter_Value"])))
        rows.sort()
        return rows
    F, Wr = gemm_rows(fetch_csv), gemm_rows(write_csv_path)
    ndisp = lambda r: 0 if r["persist"] == 2 else (2 if r["split_k"] > 1 else 1)          # noqa: E731  (pair member 0: no dispatch of its own)
    per_step = sum(ndisp(r) for r in rec)
    # the profiled command runs 2 warm-up steps + the instrumented one: keep the dispatches of the LAST step
    for rows in (F, Wr):
        assert len(rows) >= per_step, (len(rows), per_step)          # the first warm-up step also builds cached constant tables
    F, Wr = F[-per_step:], Wr[-per_step:]
    i = 0
    for r in rec:
        n = ndisp(r)
        assert n == 0 or ("splitk_reduce" not in F[i][1] and (n == 1 or "splitk_reduce" in F[i + 1][1])), (i, F[i], r)
        r["fetch_bytes"] = sum(v for _, _, v in F[i:i + n]) * 1024 * 2          # gfx950: FETCH_SIZE tallies wide reads at half size
        r["write_bytes"] = sum(v for _, _, v in Wr[i:i + n]) * 1024
        i += n
    write_csv(out_csv, rec)
    tot = dict(fetch_bytes_per_step=sum(r["fetch_bytes"] for r in rec), write_bytes_per_step=sum(r["write_bytes"] for r in rec),
               algorithmic_bytes_per_step=sum(r["alg_bytes"] for r in rec), launches_per_step=len(rec),
               weight_bytes_per_step=sum(r.get("w_bytes", 0.0) for r in rec),
               algorithmic_bytes_per_step_8xcd=sum(r["alg_bytes"] + 7 * r.get("w_bytes", 0.0) for r in rec),
               note_8xcd="FETCH_SIZE counts L2 misses; the 8 XCDs have private L2s, so a weight panel is fetched once per XCD: the floor of this "
                         "counter is A + C + 8 W (algorithmic_bytes_per_step_8xcd), most of the 7 extra copies being served by the Infinity Cache",
               kernel="conv_gemm_dma_kernel + conv_gemm_kernel + rowstream / rowchain128 / rowmlp128 + skinny/narrow variants + split-K reducers (all launches of the family in one step)",
               corrections="FETCH_SIZE KiB x2 (MI355X_MICROARCH.md HBM section), WRITE_SIZE KiB x1",
               source="rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/gemm_shapes_csv.py x.csv --profile-only")
    json.dump(tot, open(out_csv.replace("gemm_shapes.csv", "traffic.json"), "w"), indent=1)
    print(json.dumps(tot, indent=1))


if __name__ == "__main__":
    out = sys.argv[1]
    if "--join" in sys.argv:
        i = sys.argv.index("--join")
        join(out, *sys.argv[i + 1:i + 4])
    else:
        lj = sys.argv[sys.argv.index("--launches") + 1] if "--launches" in sys.argv else None
        timing_pass(out, lj, profile_only="--profile-only" in sys.argv)

"""Stand-alone probe of the exact-split bf16 contraction (csrc/gemm_split3.h) against the exact fp32-MFMA kernel -- VERDICT r5 item 1 (a).

Three shapes of the path, operands PRE-SPLIT into planes (st_split3_pack, outside the timed region):
  conv1x5   8 192 x 256 x 1 920   SepConvGRU's 1x5 z|r convolution (gru.py:44-59), two passes of a 64 x 64 map, Cin = 384
  aggregate 2 x (4 096 x 128 x 4 096)   GMA aggregate attn @ v (gma.py:102-115), one per pass
  corr      8 x (4 096 x 4 096 x 256)   all-pairs correlation volume at batch 8 (encoder.py:359-369)
For each: time of both kernels (HIP events over `iters` launches), fp32-equivalent TFLOP/s = 2 M N K / t, error of both results against the
fp64 product (max and rms, relative to the rms of the result), and the +-inf / NaN / subnormal propagation check.
Gate: >= 1.5 x the exact kernel on all three, error <= 1.25 x the exact kernel's, specials propagate alike.

    python tools/split3_probe.py [--iters 50] [--tiles 31,32,33,34] [--json out.json]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stitch_amd  # noqa: E402

ops = stitch_amd.ops
lib = ops.lib


def pack(x2d, chunk_rows=None):
    """fp32 [rows, C] -> planes tensor bf16 [3, C/32, chunk_rows, 32] (st_split3_pack)"""
    rows, Cc = x2d.shape
    chunk_rows = chunk_rows or rows
    planes = torch.empty((3, Cc // 32, chunk_rows, 32), device=x2d.device, dtype=torch.bfloat16)
    ops.check(lib.st_split3_pack(C.c_void_p(x2d.data_ptr()), C.c_void_p(planes.data_ptr()), rows, Cc, x2d.stride(0), planes.stride(0), chunk_rows,
                                 ops._stream()), "st_split3_pack")
    return planes


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3          # us


def err(c, ref64):
    d = (c.double() - ref64)
    scale = ref64.pow(2).mean().sqrt().item()
    return d.abs().max().item() / scale, d.pow(2).mean().sqrt().item() / scale


def split3_desc(ap, wp, out, *, geom, Cin, N, batch=1, bsa=0, bsw=0, bsc=0, tile=0, split_k=0, a_rows=None, w_rows=None, bias=None):
    d = ops.GemmDesc()
    B, H, W, kh, kw, sh, sw, ph, pw = geom
    Ho = (H + 2 * ph - (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - (kw - 1) - 1) // sw + 1
    d.a, d.w, d.c = ap.data_ptr(), wp.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.M, d.N, d.K = B * Ho * Wo, N, kh * kw * Cin
    d.H, d.W, d.Cin, d.ldx = H, W, Cin, Cin
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.Ho, d.Wo = kh, kw, sh, sw, ph, pw, Ho, Wo
    d.ldw, d.ldc = d.K, out.stride(-2)
    d.alpha = 1.0
    d.batch, d.batch_stride_a, d.batch_stride_w, d.batch_stride_c = batch, bsa, bsw, bsc
    d.tile_cfg, d.split_k = tile, split_k
    d.split3 = 1
    d.a_plane_stride, d.w_plane_stride = ap.stride(0), wp.stride(0)
    d.a_rows = a_rows if a_rows is not None else ap.shape[2]
    d.w_rows = w_rows if w_rows is not None else wp.shape[2]
    if batch <= 1 and split_k != 1:
        ws = ops._workspace(out.device)
        d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
    return d


def launch(d):
    ops.check(lib.st_conv_gemm(C.byref(d), ops._stream()), "st_conv_gemm(split3)")


def plan():
    p = (C.c_int32 * 4)()
    lib.st_gemm_last_plan(p)
    return list(p)


def run_shape(name, make, iters, tiles):
    r = make()
    flop = r["flop"]
    t_exact = timeit(r["exact"], iters)
    r["exact"]()
    torch.cuda.synchronize()
    e_exact = err(r["out_exact"], r["ref64"])
    res = dict(shape=name, gflop=flop / 1e9, exact_us=t_exact, exact_tflops=flop / t_exact / 1e6, exact_err_max=e_exact[0], exact_err_rms=e_exact[1], split3={})
    best = None
    for tile in tiles:
        try:
            fn = r["split3"](tile)
            fn()
            torch.cuda.synchronize()
        except Exception as ex:       # rejected configuration
            res["split3"][tile] = dict(error=str(ex))
            continue
        pl = plan()
        e3 = err(r["out_split3"], r["ref64"])
        t3 = timeit(fn, iters)
        row = dict(us=t3, tflops_fp32_equiv=flop / t3 / 1e6, speedup=t_exact / t3, err_max=e3[0], err_rms=e3[1], err_rms_ratio=e3[1] / e_exact[1],
                   err_max_ratio=e3[0] / e_exact[0], split_k=pl[2])
        res["split3"][tile] = row
        if best is None or t3 < best[1]:
            best = (tile, t3)
        print(f"  {name} tile {tile}: {t3:8.1f} us  {row['tflops_fp32_equiv']:6.1f} TF  x{row['speedup']:.2f}  err rms {e3[1]:.3e} (exact {e_exact[1]:.3e}, ratio {row['err_rms_ratio']:.2f})"
              f"  max {e3[0]:.3e} (ratio {row['err_max_ratio']:.2f})  split_k {pl[2]}", flush=True)
    res["best_tile"] = best[0] if best else None
    res["best_speedup"] = t_exact / best[1] if best else None
    print(f"{name}: exact {t_exact:.1f} us {res['exact_tflops']:.1f} TF; best split3 tile {res['best_tile']} x{res['best_speedup']:.2f}", flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--tiles", default="31,32,33,34,37")
    ap.add_argument("--json", default="")
    ap.add_argument("--shapes", default="conv1x5,aggregate,corr")
    a = ap.parse_args()
    tiles = [int(t) for t in a.tiles.split(",")]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1234)

    def conv(N, Cin, kh, kw):
        def make():
            B, H, W = 2, 64, 64
            x = torch.randn(B * H * W, Cin, generator=g).to(dev)
            w = (torch.randn(N, kh * kw * Cin, generator=g) / (kh * kw * Cin) ** 0.5).to(dev)
            geom = (B, H, W, kh, kw, 1, 1, kh // 2, kw // 2)
            out_e = torch.empty(B * H * W, N, device=dev)
            out_s = torch.empty_like(out_e)
            xp, wp = pack(x), pack(w)        # W [N, K]: chunk k/32 of row n -> [K/32][N][32]
            # fp64 reference: the conv as kh*kw shifted products
            x64 = x.double().view(B, H, W, Cin)
            ref = torch.zeros(B, H, W, N, dtype=torch.float64, device=dev)
            w64 = w.double().view(N, kh, kw, Cin)
            for ky in range(kh):
                for kx in range(kw):
                    dy, dx = ky - kh // 2, kx - kw // 2
                    ylo, yhi = max(0, -dy), min(H, H - dy)
                    xlo, xhi = max(0, -dx), min(W, W - dx)
                    ref[:, ylo:yhi, xlo:xhi] += x64[:, ylo + dy:yhi + dy, xlo + dx:xhi + dx] @ w64[:, ky, kx].t()
            return dict(flop=2.0 * B * H * W * N * kh * kw * Cin, ref64=ref.view(-1, N), out_exact=out_e, out_split3=out_s,
                        exact=lambda: ops.conv_gemm(x, w, out_e, geom=geom),
                        split3=lambda tile: (lambda d=split3_desc(xp, wp, out_s, geom=geom, Cin=Cin, N=N, tile=tile): launch(d)))
        return make

    def aggregate():
        Bp, M, N, K = 2, 4096, 128, 4096
        attn = torch.softmax(torch.randn(Bp, M, K, generator=g) * 2.0, dim=-1).to(dev)
        vt = torch.randn(Bp, N, K, generator=g).to(dev)          # v^T: [channels, pixels]
        out_e = torch.empty(Bp, M, N, device=dev)
        out_s = torch.empty_like(out_e)
        ap_ = pack(attn.view(Bp * M, K), Bp * M)                  # [3][K/32][Bp*M][32]: batch b = rows b*M ..
        wp = pack(vt.view(Bp * N, K), Bp * N)
        ref = attn.double() @ vt.double().transpose(1, 2)
        geom = (1, 1, M, 1, 1, 1, 1, 0, 0)

        def exact():
            ops.conv_gemm(attn.view(Bp * M, K), vt.view(Bp * N, K)[:N], out_e.view(Bp * M, N), M=M, batch=Bp, bsa=M * K, bsw=N * K, bsc=M * N)

        def split3(tile):
            d = split3_desc(ap_, wp, out_s, geom=geom, Cin=K, N=N, batch=Bp, bsa=M * 32, bsw=N * 32, bsc=M * N, tile=tile, a_rows=Bp * M, w_rows=Bp * N)
            return lambda: launch(d)
        return dict(flop=2.0 * Bp * M * N * K, ref64=ref.view(-1, N), out_exact=out_e.view(-1, N), out_split3=out_s.view(-1, N), exact=exact, split3=split3)

    def corr():
        Bp, Np, Cc = 8, 4096, 256
        f1 = torch.randn(Bp, Np, Cc, generator=g).to(dev)
        f2 = torch.randn(Bp, Np, Cc, generator=g).to(dev)
        out_e = torch.empty(Bp, Np, Np, device=dev)
        out_s = torch.empty_like(out_e)
        p1, p2 = pack(f1.view(Bp * Np, Cc), Bp * Np), pack(f2.view(Bp * Np, Cc), Bp * Np)
        ref = (f1[:2].double() @ f2[:2].double().transpose(1, 2))      # fp64 check on two of the eight batches
        geom = (1, 1, Np, 1, 1, 1, 1, 0, 0)

        def split3(tile):
            d = split3_desc(p1, p2, out_s, geom=geom, Cin=Cc, N=Np, batch=Bp, bsa=Np * 32, bsw=Np * 32, bsc=Np * Np, tile=tile, a_rows=Bp * Np, w_rows=Bp * Np)
            return lambda: launch(d)
        return dict(flop=2.0 * Bp * Np * Np * Cc, ref64=ref.view(-1, Np), out_exact=out_e[:2].reshape(-1, Np), out_split3=out_s[:2].reshape(-1, Np),
                    exact=lambda: ops.corr_volume(f1, f2, out_e), split3=split3)

    makers = dict(conv1x5=conv(256, 384, 1, 5), aggregate=aggregate, corr=corr,
                  # the other decoder convolutions of one refinement iteration (gru.py:44-59,246-254,5-13), M = 8 192
                  zr5x1=conv(256, 384, 5, 1), q1x5=conv(128, 384, 1, 5), convc2=conv(192, 256, 3, 3), convf2=conv(64, 128, 3, 3),
                  conv3x3=conv(126, 256, 3, 3), fh1=conv(256, 128, 3, 3),
                  # duration vs K at fixed M x N (fixed cost per launch / in-loop rate): 1x5 conv, Cin = 32 .. 768
                  k160=conv(256, 32, 1, 5), k640=conv(256, 128, 1, 5), k3840=conv(256, 768, 1, 5))
    results = []
    for name in a.shapes.split(","):
        results.append(run_shape(name, makers[name], a.iters, tiles))
        torch.cuda.empty_cache()

    # ---- special values: +-inf, NaN and subnormal inputs must propagate as the exact kernel's do
    M, N, K = 256, 128, 256
    x = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g).to(dev)
    x[3, 7] = float("inf"); x[5, 9] = float("-inf"); x[8, 1] = float("nan"); x[11, 2] = 1e-40; x[12, :] = 0; x[12, 5] = 3e-39
    w[4, 7] = 0.0                                   # inf * 0 -> NaN in both
    w[6, 30] = float("inf")
    x[20, 40] = 3.4e38                              # finite, rounds to inf in bf16: must stay finite
    w[:, 40] = w[:, 40].clamp(-0.4, 0.4)
    oe, os_ = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.conv_gemm(x, w, oe)
    d = split3_desc(pack(x), pack(w), os_, geom=(1, 1, M, 1, 1, 1, 1, 0, 0), Cin=K, N=N, tile=34, split_k=1)
    launch(d)
    torch.cuda.synchronize()
    # an inf operand meets the other operand's mid / lo parts (zero, or of the opposite sign) in separate products: inf * 0 or inf - inf,
    # so where the fp32 chain gives +-inf the split gives NaN.  What must hold: non-finite in -> non-finite out at exactly the same
    # outputs, every NaN of the exact kernel is a NaN here, and finite inputs that only bf16 rounding would push to inf stay finite.
    nonfinite_equal = bool((torch.isfinite(oe) == torch.isfinite(os_)).all())
    nan_superset = bool((torch.isnan(os_) | ~torch.isnan(oe)).all())
    inf_to_nan = int((torch.isinf(oe) & torch.isnan(os_)).sum())
    same_nan, same_inf = nonfinite_equal, nan_superset
    fin = torch.isfinite(oe) & torch.isfinite(os_)
    keep = torch.ones(N, dtype=torch.bool, device=dev); keep[6] = False
    sub_rel = ((oe[12][keep] - os_[12][keep]).abs().max() / oe[12][keep].abs().max().clamp_min(1e-45)).item()
    specials = dict(nonfinite_pattern_equal=nonfinite_equal, exact_nans_are_nans=nan_superset, inf_outputs_that_became_nan=inf_to_nan,
                    finite_max_rel_diff=((oe[fin] - os_[fin]).abs() / oe[fin].abs().clamp_min(1e-30)).max().item(),
                    big_row_finite=bool(torch.isfinite(os_[20]).all()), big_row_rel_diff=((oe[20] - os_[20]).abs().max() / oe[20].abs().max()).item(),
                    subnormal_row_exact_absmax=oe[12][keep].abs().max().item(), subnormal_row_split3_absmax=os_[12][keep].abs().max().item(),
                    subnormal_row_rel_diff=sub_rel, nan_count_exact=int(torch.isnan(oe).sum()), inf_count_exact=int(torch.isinf(oe).sum()))
    print("specials:", specials, flush=True)

    gate = all(r["shape"] not in ("conv1x5", "aggregate", "corr") or r["best_speedup"] and r["best_speedup"] >= 1.5 and r["split3"][r["best_tile"]]["err_rms_ratio"] <= 1.25 for r in results) and same_nan and same_inf
    out = dict(results=results, specials=specials, gate_speedup_1p5_on_all=gate, device=torch.cuda.get_device_name(0))
    print(json.dumps(out))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

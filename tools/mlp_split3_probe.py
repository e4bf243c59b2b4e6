"""Stand-alone probe of st_mlp128_split3 against st_mlp128 (GPU box): duration (hipGraph of N launches), error of both against fp64.
    python tools/mlp_split3_probe.py [--iters 40]"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stitch_amd  # noqa: E402

ops = stitch_amd.ops


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    gen = torch.Generator().manual_seed(3)
    out = []
    for M, proj in ((65536, True), (65536, False), (32768, True), (32768, False), (8192, True)):
        hidden = 512
        x, x0 = torch.randn(M, 128, generator=gen).cuda(), torch.randn(M, 128, generator=gen).cuda()
        w1, b1 = (torch.randn(hidden, 128, generator=gen) / 128 ** 0.5).cuda(), (torch.randn(hidden, generator=gen) * 0.1).cuda()
        w2, b2 = (torch.randn(128, hidden, generator=gen) / hidden ** 0.5).cuda(), (torch.randn(128, generator=gen) * 0.1).cuda()
        wp, bp = (torch.randn(128, 128, generator=gen) / 128 ** 0.5).cuda(), (torch.randn(128, generator=gen) * 0.1).cuda()
        pj = (wp, bp, x0) if proj else None
        img = ops.mlp128_split3_pack(w1, b1, w2, proj=(wp, bp) if proj else None)
        oe, os_ = torch.empty(M, 128, device="cuda"), torch.empty(M, 128, device="cuda")
        te = timed(lambda: ops.mlp128(x, oe, w1, b1, w2, b2, ln_eps=1e-6, proj=pj), a.iters)
        ts = timed(lambda: ops.mlp128(x, os_, w1, b1, w2, b2, ln_eps=1e-6, proj=pj, image=img), a.iters)
        xd = x.double()
        if proj:
            xd = F.linear(xd, wp.double(), bp.double()) + x0.double()
        ref = F.linear(F.gelu(F.linear(F.layer_norm(xd, (128,), None, None, 1e-6), w1.double(), b1.double())), w2.double(), b2.double()) + xd
        sc = ref.abs().max().item()
        ee, es = (oe.double() - ref).pow(2).mean().sqrt().item() / sc, (os_.double() - ref).pow(2).mean().sqrt().item() / sc
        fl = 2.0 * M * 128 * (2 * hidden + (128 if proj else 0))
        rec = dict(M=M, proj=proj, exact_us=te, split3_us=ts, speedup=te / ts, exact_tflops=fl / te * 1e-6, split3_tflops_fp32_equiv=fl / ts * 1e-6,
                   split3_bf16_tflops=6 * fl / ts * 1e-6, exact_rms_vs_fp64=ee, split3_rms_vs_fp64=es, nonfinite=int((~torch.isfinite(os_)).sum().item()))
        out.append(rec)
        print(f"M {M:6d} proj {int(proj)}: exact {te:7.1f} us ({rec['exact_tflops']:6.1f} TF)  split3 {ts:7.1f} us ({rec['split3_tflops_fp32_equiv']:6.1f} TF fp32-equiv, "
              f"{rec['split3_bf16_tflops']:6.1f} TF bf16)  x{te / ts:4.2f}   rms vs fp64: exact {ee:.3e} split3 {es:.3e} (ratio {es / ee:4.2f})  non-finite {rec['nonfinite']}", flush=True)
    # LayerNorm -> Linear(128 -> N): st_conv_gemm(a_ln) (rowstream_gemm_kernel) against st_rowlin128_split3
    for M, N in ((65536, 384), (32768, 384), (65536, 128)):
        x = torch.randn(M, 128, generator=gen).cuda()
        w, b = (torch.randn(N, 128, generator=gen) / 128 ** 0.5).cuda(), (torch.randn(N, generator=gen) * 0.1).cuda()
        img = ops.rowlin128_split3_pack(w, b)
        oe, os_ = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        te = timed(lambda: ops.conv_gemm(x, w, oe, bias=b, ln_eps=1e-5), a.iters)
        ts = timed(lambda: ops.rowlin128_split3(x, os_, img, ln_eps=1e-5), a.iters)
        ref = F.linear(F.layer_norm(x.double(), (128,), None, None, 1e-5), w.double(), b.double())
        sc = ref.pow(2).mean().sqrt().item()
        ee, es = (oe.double() - ref).pow(2).mean().sqrt().item() / sc, (os_.double() - ref).pow(2).mean().sqrt().item() / sc
        out.append(dict(op="ln_linear", M=M, N=N, exact_us=te, split3_us=ts, speedup=te / ts, exact_rms_vs_fp64=ee, split3_rms_vs_fp64=es))
        print(f"LN + Linear M {M:6d} N {N:3d}: exact {te:7.1f} us  split3 {ts:7.1f} us  x{te / ts:4.2f}   rms vs fp64: exact {ee:.3e} split3 {es:.3e} (ratio {es / ee:4.2f})", flush=True)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

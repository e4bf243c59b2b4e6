"""Compare the register-staged (cfg 3) and LDS-DMA pipelined (cfg 12/13) GEMM kernels on the path's shapes."""
import sys, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
torch.manual_seed(0)
# (label, B,H,W, Cin, kh,kw, N)
SHAPES = [("gru zr 1x5", 2, 64, 64, 384, 1, 5, 256), ("gru q 5x1", 2, 64, 64, 384, 5, 1, 128),
          ("convc2 3x3", 2, 64, 64, 256, 3, 3, 192), ("conv 3x3", 2, 64, 64, 256, 3, 3, 126),
          ("fh1 3x3", 2, 64, 64, 128, 3, 3, 256), ("agg 1x1 K4096", 1, 1, 4096, 4096, 1, 1, 128),
          ("vert 1x1 K128", 1, 1, 65536, 128, 1, 1, 128), ("mlp fc1", 1, 1, 65536, 128, 1, 1, 512),
          ("mlp fc2", 1, 1, 65536, 512, 1, 1, 128), ("twins 1x1", 1, 1, 8192, 256, 1, 1, 1024),
          ("twins fc2", 1, 1, 8192, 1024, 1, 1, 256), ("pe c4 6x6s2", 64, 16, 16, 32, 6, 6, 64),
          ("res 3x3 2048", 2, 32, 32, 256, 3, 3, 256), ("corr", 1, 1, 4096, 256, 1, 1, 4096)]
def run(x, w, out, geom, tile, split=0, iters=20):
    for _ in range(3): ops.conv_gemm(x, w, out, geom=geom, tile=tile, split_k=split)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.conv_gemm(x, w, out, geom=geom, tile=tile, split_k=split)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for label, B, H, W, Cin, kh, kw, N in SHAPES:
    stride = 2 if "s2" in label else 1
    pad_h, pad_w = (kh // 2, kw // 2) if stride == 1 else (2, 2)
    x = torch.randn(B * H * W, Cin, device="cuda")
    w = torch.randn(N, kh * kw * Cin, device="cuda") / (kh * kw * Cin) ** 0.5
    geom = (B, H, W, kh, kw, stride, stride, pad_h, pad_w)
    Ho, Wo = (H + 2 * pad_h - kh) // stride + 1, (W + 2 * pad_w - kw) // stride + 1
    M = B * Ho * Wo
    ref = torch.empty(M, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    t3 = run(x, w, ref, geom, 3)
    fl = 2.0 * M * N * kh * kw * Cin
    line = f"{label:>14} M={M:>6} N={N:>4} K={kh*kw*Cin:>5}  reg64x64 {t3:7.1f}us {fl/t3/1e6:6.1f}TF |"
    for tile in (13, 12):
        out.zero_()
        t = run(x, w, out, geom, tile)
        err = (out - ref).abs().max().item()
        line += f" dma{tile} {t:7.1f}us {fl/t/1e6:6.1f}TF err {err:.1e} |"
    for tile, split in ((13, 1), (13, 2), (12, 2)):
        out.zero_()
        try:
            t = run(x, w, out, geom, tile, split)
        except Exception:
            continue
        err = (out - ref).abs().max().item()
        line += f" d{tile}/s{split} {t:6.1f}us err {err:.1e}|"
    print(line, flush=True)

# persistent-M path: ragged M, bias + residual epilogue, column-slice operands
for M, N, K in ((65000, 128, 128), (40001, 384, 256), (70000, 128, 512)):
    xw = torch.randn(M, K + 64, device="cuda"); x = xw[:, 32:32 + K]
    w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
    ref = torch.empty(M, N, device="cuda"); out = torch.empty(M, N + 32, device="cuda")[:, :N]
    ops.conv_gemm(x, w, ref, bias=b, act="gelu", epi="add", aux1=res, tile=3)
    ops.conv_gemm(x, w, out, bias=b, act="gelu", epi="add", aux1=res, tile=13)
    torch.cuda.synchronize()
    print(f"persist check M={M} N={N} K={K}: max err {(out - ref).abs().max().item():.2e}")

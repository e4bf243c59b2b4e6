import sys, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
torch.manual_seed(0)
def run(x, w, out, geom, tile, split=1, iters=20, precision=0):
    for _ in range(3): ops.conv_gemm(x, w, out, geom=geom, tile=tile, split_k=split, precision=precision)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.conv_gemm(x, w, out, geom=geom, tile=tile, split_k=split, precision=precision)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for label, B, H, W, Cin, kh, kw, N in [("gru q 5x1", 2, 64, 64, 384, 5, 1, 128), ("gru zr 1x5", 2, 64, 64, 384, 1, 5, 256), ("1x1 K2048", 1, 1, 8192, 2048, 1, 1, 128),
                                       ("big 1x1", 1, 1, 65536, 2048, 1, 1, 256)]:
    x = torch.randn(B * H * W, Cin, device="cuda"); w = torch.randn(N, kh * kw * Cin, device="cuda") * 0.02
    geom = (B, H, W, kh, kw, 1, 1, kh // 2, kw // 2); M = B * H * W
    out = torch.empty(M, N, device="cuda"); fl = 2.0 * M * N * kh * kw * Cin
    line = f"{label:>12}"
    for tile in (3, 15, 13, 14, 16):
        t = run(x, w, out, geom, tile); line += f" | cfg{tile} {t:6.1f}us {fl/t/1e6:6.1f}TF"
    for tile in (13, 16):
        t = run(x, w, out, geom, tile, precision=100); line += f" | cfg{tile}/nomem {t:6.1f}us {fl/t/1e6:6.1f}TF"
    print(line, flush=True)

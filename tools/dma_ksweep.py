import sys, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
def run(x, w, out, tile, iters=30, precision=0):
    for _ in range(3): ops.conv_gemm(x, w, out, tile=tile, split_k=1, precision=precision)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.conv_gemm(x, w, out, tile=tile, split_k=1, precision=precision)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N in ((8192, 128), (8192, 256), (16384, 256)):
    for K in (128, 384, 768, 1920, 3840, 7680):
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02
        out = torch.empty(M, N, device="cuda")
        line = f"M={M} N={N} K={K:>5} tiles/K={K//32:>4}"
        for tile, prec in ((3, 0), (13, 0), (13, 100)):
            t = run(x, w, out, tile, precision=prec)
            line += f" | cfg{tile}{'/nomem' if prec else ''} {t:7.1f}us {t*1e3/(K//32):6.0f}ns/tile"
        print(line, flush=True)

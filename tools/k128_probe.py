import sys, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
def run(x, w, out, iters=30, **kw):
    for _ in range(3): ops.conv_gemm(x, w, out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.conv_gemm(x, w, out, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N, K in ((65536, 128, 128), (65536, 512, 128), (65536, 128, 512), (65536, 384, 128)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; out = torch.empty(M, N, device="cuda")
    print(f"M={M} N={N} K={K}: real {run(x, w, out):6.1f} us | no operand reads {run(x, w, out, precision=100):6.1f} us | no reads, no stores {run(x, w, out, precision=101):6.1f} us | mfma floor {2.0*M*N*K/157.3e6:5.1f} us")

import sys, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
M, N, K = 65536, 512, 128
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
def run(**kw):
    for _ in range(3): ops.conv_gemm(x, w, out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_gemm(x, w, out, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for name, kw in [("plain", {}), ("bias", dict(bias=b)), ("bias+relu", dict(bias=b, act="relu")), ("bias+gelu", dict(bias=b, act="gelu")),
                 ("bias+tanh", dict(bias=b, act="tanh")), ("bias+sigmoid", dict(bias=b, act="sigmoid"))]:
    print(f"{name:>14}: {run(**kw):7.1f} us")

"""Which feature of the SepConvGRU q convolution (8192 x 128 x 1920 on planes) costs what: back-to-back launches of the same split3 GEMM with the
second A source, the GRU epilogue (3 fp32 operands + tanh), the plane emission and the fp32 store switched on one by one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stitch_amd
ops = stitch_amd.ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
B, H, W = 2, 64, 64
R = B * H * W
hx = torch.randn(R, 384, generator=g).to(dev)
hA, hB = ops.split3_pack(hx), ops.split3_pack(torch.randn(R, 384, generator=g).to(dev))
w = ops.split3_pack((torch.randn(128, 5 * 384, generator=g) * 0.02).to(dev))
tab = (torch.randn(R, 384, generator=g) * 0.3).to(dev)
z = torch.rand(R, 128, generator=g).to(dev)
out = torch.empty(R, 128, device=dev)
geom = (B, H, W, 1, 5, 1, 1, 0, 2)

def t(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
cases = {
    "plain store": dict(),
    "+ second A source": dict(a2=hB, a2_channels=128),
    "+ aux0 table + tanh": dict(a2=hB, a2_channels=128, aux0=tab[:, 256:], act="tanh"),
    "+ GRU blend (z, h operands)": dict(a2=hB, a2_channels=128, aux0=tab[:, 256:], act="tanh", epi="gru", aux1=z, aux2=hx[:, :128]),
    "+ planes out": dict(a2=hB, a2_channels=128, aux0=tab[:, 256:], act="tanh", epi="gru", aux1=z, aux2=hx[:, :128], out_planes=hA.cols(0, 128)),
    "planes out only (plain store + planes)": dict(out_planes=hA.cols(0, 128)),
    "planes out, no fp32 store": dict(out_planes=hA.cols(0, 128), no_f32=True),
}
for name, kw in cases.items():
    print(f"{t(lambda: ops.conv_gemm(hA, w, out, geom=geom, **kw)):7.1f} us  {name}", flush=True)
# the fp32 kernel with the full epilogue, for reference
wf = (torch.randn(128, 5 * 384, generator=g) * 0.02).to(dev)
hxB = torch.randn(R, 384, generator=g).to(dev)
print(f"{t(lambda: ops.conv_gemm(hx, wf, out, geom=geom, a2=hxB, a2_channels=128, aux0=tab[:, 256:], act='tanh', epi='gru', aux1=z, aux2=hx[:, :128])):7.1f} us  fp32-MFMA kernel, full q epilogue")

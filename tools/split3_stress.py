"""Determinism stress of the split3 kernels under concurrency: the same launches on 3 streams at once, repeated, every result compared bit for
bit with the result of the same launch run alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stitch_amd
ops = stitch_amd.ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
B, H, W = 2, 64, 64
R = B * H * W
cases = []
for (N, Cin, kh, kw, kwargs) in [(256, 384, 1, 5, {}), (128, 384, 5, 1, {}), (126, 256, 3, 3, {}), (256, 128, 3, 3, dict(tile=34)), (192, 256, 3, 3, dict(tile=32))]:
    x = torch.randn(R, Cin, generator=g).to(dev)
    w = (torch.randn(N, kh * kw * Cin, generator=g) / (kh * kw * Cin) ** 0.5).to(dev)
    cases.append((ops.split3_pack(x), ops.split3_pack(w), N, (B, H, W, kh, kw, 1, 1, kh // 2, kw // 2), kwargs))
streams = [torch.cuda.Stream() for _ in range(3)]
wss = [ops.new_workspace(dev) for _ in range(3)]
outs = [[torch.empty(R, c[2], device=dev) for c in cases] for _ in range(3)]
planes = [[ops.Planes(R, 256, dev) for c in cases] for _ in range(3)]
ref, refp = [], []
for i, (xp, wp, N, geom, kw) in enumerate(cases):
    o = torch.empty(R, N, device=dev); p = ops.Planes(R, 256, dev); p.t.zero_()
    ops.conv_gemm(xp, wp, o, geom=geom, act="relu", out_planes=p.cols(0, (N + 31) // 32 * 32), **kw)
    torch.cuda.synchronize()
    ref.append(o.clone()); refp.append(p.t.clone())
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    for s in range(3):
        with torch.cuda.stream(streams[s]), ops.workspace_scope(wss[s]):
            for i, (xp, wp, N, geom, kw) in enumerate(cases):
                planes[s][i].t.zero_()
                ops.conv_gemm(xp, wp, outs[s][i], geom=geom, act="relu", out_planes=planes[s][i].cols(0, (N + 31) // 32 * 32), **kw)
    torch.cuda.synchronize()
    for s in range(3):
        for i in range(len(cases)):
            if not torch.equal(outs[s][i], ref[i]) or not torch.equal(planes[s][i].t, refp[i]):
                bad += 1
                d = (outs[s][i] - ref[i]).abs()
                print(f"iter {it} stream {s} case {i}: {int((d > 0).sum())} elements differ, max {d.max().item():.3e}; planes differ {int((planes[s][i].t != refp[i]).sum())}", flush=True)
print("mismatching results:", bad)

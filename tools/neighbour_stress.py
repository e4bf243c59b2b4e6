"""Does a kernel corrupt the loads of OTHER kernels that run beside it?  Stream B loops an aggressor (the split3 conv, or the fp32-MFMA conv as the
control); stream A loops victims (the path's bilinear resize and a plain torch elementwise kernel) on constant inputs and compares every result
with the one computed on an idle chip.    python tools/neighbour_stress.py split3|exact [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stitch_amd
ops = stitch_amd.ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "split3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = torch.Generator().manual_seed(11)
B, H, W, Cin, N = 2, 64, 64, 256, 192
x = torch.randn(B * H * W, Cin, generator=g).to(dev)
w = (torch.randn(N, 9 * Cin, generator=g) / 48).to(dev)
out = torch.empty(B * H * W, N, device=dev)
geom = (B, H, W, 3, 3, 1, 1, 1, 1)
xp, wp = ops.split3_pack(x), ops.split3_pack(w)
flow = (torch.randn(2, 2, 512, 512, generator=g) * 5).to(dev)
src = torch.randn(8 * 1024 * 1024, generator=g).to(dev)
ref_r = ops.resize_bilinear(flow, 320, 416, True, div=(512 / 416.0, 512 / 320.0)).clone()
ref_e = (src * 2.0 + 1.0).clone()
idx = torch.randint(0, src.numel(), (4 * 1024 * 1024,), generator=g).to(dev)
small = src[:4 * 1024 * 1024]
VICT = {"div": lambda: 3.0 / small, "exp": lambda: torch.exp(small * 0.1), "cvt_int": lambda: (small * 100.0).to(torch.int32), "sqrt": lambda: torch.sqrt(small.abs()),
        "gather": lambda: src[idx], "fdiv2": lambda: small / 1.23, "floor": lambda: torch.floor(small * 7.3), "mul_int": lambda: (idx * 3 + 1), "idiv": lambda: idx // 416}
ref_v = {k: f().clone() for k, f in VICT.items()}
bad_v = {k: 0 for k in VICT}
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ws = ops.new_workspace(dev)
bad_r = bad_e = 0
for it in range(iters):
    with torch.cuda.stream(sb), ops.workspace_scope(ws):
        for _ in range(12):
            if which == "split3":
                ops.conv_gemm(xp, wp, out, geom=geom, act="relu")
            else:
                ops.conv_gemm(x, w, out, geom=geom, act="relu")
    with torch.cuda.stream(sa):
        rs = [ops.resize_bilinear(flow, 320, 416, True, div=(512 / 416.0, 512 / 320.0)) for _ in range(12)]
        es = [src * 2.0 + 1.0 for _ in range(4)]
        vs = {k: [f() for _ in range(3)] for k, f in VICT.items()}
    torch.cuda.synchronize()
    for k, lst in vs.items():
        for t in lst:
            if not torch.equal(t, ref_v[k]):
                bad_v[k] += 1
                if bad_v[k] <= 2:
                    ix = (t != ref_v[k]).nonzero().flatten()
                    print(f"iter {it}: victim {k} differs in {ix.numel()} elements; lanes (mod 64): {sorted(set((ix % 64).tolist()))[:20]}", flush=True)
    for r in rs:
        if not torch.equal(r, ref_r):
            bad_r += 1
            if bad_r <= 3:
                ix = (r != ref_r).nonzero()
                print(f"iter {it}: resize differs in {ix.shape[0]} elements; cols {ix[:, 3].min().item()}..{ix[:, 3].max().item()} rows {ix[:, 2].unique().tolist()[:4]}", flush=True)
    for e in es:
        if not torch.equal(e, ref_e):
            bad_e += 1
            if bad_e <= 3:
                ix = (e != ref_e).nonzero()
                print(f"iter {it}: elementwise differs in {ix.shape[0]} elements; first {ix[0].item()} (mod 64 = {ix[0].item() % 64})", flush=True)
print("victims:", bad_v, "of", iters * 3, "each")
print(f"aggressor {which}: corrupted resize results {bad_r} of {iters * 12}, corrupted elementwise results {bad_e} of {iters * 4}")

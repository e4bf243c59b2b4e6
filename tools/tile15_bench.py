"""64x128 tile (tile_cfg 15: one workgroup per CU, two accumulator tiles per wave) against the default 64x64 on the decoder's conv shapes."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
ops = stitch_amd.ops
def run(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, H, W = 2, 64, 64
# warm the chip first: the first shapes of an earlier version of this script read 10 us slower than the same launch measured last
_x, _w, _c = torch.randn(8192, 1920, device="cuda"), torch.randn(256, 1920, device="cuda"), torch.empty(8192, 256, device="cuda")
for _ in range(40):
    run(lambda: ops.conv_gemm(_x, _w, _c, split_k=1), iters=50)
for (Cin, Co, kh, kw, ph, pw, act) in [(384, 256, 1, 5, 0, 2, "sigmoid"), (384, 256, 5, 1, 2, 0, "sigmoid"), (128, 256, 3, 3, 1, 1, "relu"), (256, 128, 3, 3, 1, 1, "relu"),
                                      (384, 128, 1, 5, 0, 2, "tanh"), (256, 256, 3, 3, 1, 1, "relu")]:
    x = torch.randn(B * H * W, Cin, device="cuda"); w = torch.randn(Co, kh * kw * Cin, device="cuda") / (kh * kw * Cin) ** 0.5
    c0, c1 = torch.empty(B * H * W, Co, device="cuda"), torch.empty(B * H * W, Co, device="cuda")
    geom = (B, H, W, kh, kw, 1, 1, ph, pw)
    t0 = run(lambda: ops.conv_gemm(x, w, c0, geom=geom, act=act, split_k=1))
    t1 = run(lambda: ops.conv_gemm(x, w, c1, geom=geom, act=act, tile=15, split_k=1))
    fl = 2.0 * B * H * W * Co * kh * kw * Cin
    print(f"conv {kh}x{kw} {Cin}->{Co}: 64x64 {t0:.1f} us ({fl / t0 / 1e6 / 157.3:.3f}) | 64x128 {t1:.1f} us ({fl / t1 / 1e6 / 157.3:.3f}) | equal {torch.equal(c0, c1)}")
# the real z|r launch of SepConvGRU (operators.hip st_sepconv_gru): ZR epilogue, pre-activation table, r*h second output
for (kh, kw, ph, pw) in [(1, 5, 0, 2), (5, 1, 2, 0)]:
    R, ld = B * H * W, 384
    hxA, hxB = torch.randn(R, ld, device="cuda"), torch.empty(R, ld, device="cuda")
    tab = torch.randn(R, 384, device="cuda"); w = torch.randn(256, 5 * ld, device="cuda") / (5 * ld) ** 0.5
    z0, z1 = torch.empty(R, 128, device="cuda"), torch.empty(R, 128, device="cuda")
    geom = (B, H, W, kh, kw, 1, 1, ph, pw)
    def f(z, tile):
        ops.conv_gemm(hxA, w, z, geom=geom, aux0=tab[:, :256], act="sigmoid", epi="zr", aux1=hxA[:, :128], out2=hxB[:, :128], tile=tile, split_k=1)
    t0, t1 = run(lambda: f(z0, 13)), run(lambda: f(z1, 15))
    print(f"z|r {kh}x{kw}: 64x64 {t0:.1f} us | 64x128 {t1:.1f} us | equal {torch.equal(z0, z1)}")

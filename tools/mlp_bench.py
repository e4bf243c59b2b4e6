"""st_mlp128 (fused LN -> fc1 + GELU -> fc2 + residual) against the two unfused launches (graph replay timing)."""
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
for M in (65536, 32768, 524288):
    x = torch.randn(M, 128, device="cuda")
    w1, b1 = torch.randn(512, 128, device="cuda") / 11, torch.randn(512, device="cuda") * 0.1
    w2, b2 = torch.randn(128, 512, device="cuda") / 22, torch.randn(128, device="cuda") * 0.1
    h, o, o2 = torch.empty(M, 512, device="cuda"), torch.empty(M, 128, device="cuda"), torch.empty(M, 128, device="cuda")
    def unfused():
        ops.conv_gemm(x, w1, h, bias=b1, act="gelu", ln_eps=1e-6)
        ops.conv_gemm(h, w2, o, bias=b2, aux0=x)
    def fused():
        ops.mlp128(x, o2, w1, b1, w2, b2, ln_eps=1e-6)
    att, x0 = torch.randn(M, 128, device="cuda"), torch.randn(M, 128, device="cuda")
    wp, bp = torch.randn(128, 128, device="cuda") / 11, torch.randn(128, device="cuda") * 0.1
    xb, o3 = torch.empty(M, 128, device="cuda"), torch.empty(M, 128, device="cuda")
    def proj_then_fused():
        ops.conv_gemm(att, wp, xb, bias=bp, aux0=x0)
        ops.mlp128(xb, o2, w1, b1, w2, b2, ln_eps=1e-6)
    def proj_fused():
        ops.mlp128(att, o3, w1, b1, w2, b2, ln_eps=1e-6, proj=(wp, bp, x0))
    tp, tpf = run(proj_then_fused), run(proj_fused)
    print(f"M={M}: projection launch + fused MLP {tp:.1f} us | projection inside the MLP launch {tpf:.1f} us | equal {torch.equal(o2, o3)}")
    tu, tf = run(unfused), run(fused)
    fl = 2.0 * M * 128 * 512 * 2
    print(f"M={M}: unfused {tu:.1f} us ({fl / tu / 1e6:.1f} TF/s) | fused {tf:.1f} us ({fl / tf / 1e6:.1f} TF/s = {fl / tf / 1e6 / 157.3:.3f} of peak) | max diff {(o - o2).abs().max().item():.2e}")

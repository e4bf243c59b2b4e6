"""st_linear_chain128 against the unfused launches (graph replay timing)."""
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
for M in (65536, 32768):
    att, x = torch.randn(M, 128, device="cuda"), torch.randn(M, 128, device="cuda")
    ws = [(torch.randn(128, 128, device="cuda") / 11, torch.randn(128, device="cuda")) for _ in range(3)]
    x1, h, o, o2 = (torch.empty(M, 128, device="cuda") for _ in range(4))
    def unfused():
        ops.conv_gemm(att, ws[0][0], x1, bias=ws[0][1], aux0=x)
        ops.conv_gemm(x1, ws[1][0], h, bias=ws[1][1], act="gelu", ln_eps=1e-5)
        ops.conv_gemm(h, ws[2][0], o, bias=ws[2][1], aux0=x1)
    def fused():
        ops.linear_chain128(att, o2, [dict(w=ws[0][0], bias=ws[0][1], res=x), dict(w=ws[1][0], bias=ws[1][1], act="gelu", ln_eps=1e-5),
                                      dict(w=ws[2][0], bias=ws[2][1], res=1)])
    def fused2():
        ops.linear_chain128(x1, o2, [dict(w=ws[1][0], bias=ws[1][1], act="gelu", ln_eps=1e-5), dict(w=ws[2][0], bias=ws[2][1], res=0)])
    tu, tf, tf2 = run(unfused), run(fused), run(fused2)
    print(f"M={M}: unfused 3 launches {tu:.1f} us | chain(3) {tf:.1f} us | chain(2: LN+f0+f3) {tf2:.1f} us | equal {torch.equal(o, o2) if False else ''}")

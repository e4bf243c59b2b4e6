"""Row-streaming kernel (tile 20) against the LDS-DMA kernel (tile 13, persistent walk) and the register-staged one
(tile 3) on the K = 64 / 128 shapes of the path; also checks that the three agree bit for bit."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
ops = stitch_amd.ops
torch.manual_seed(0)
SHAPES = [(65536, 128, 128, "none"), (65536, 128, 128, "res"), (65536, 128, 128, "gelu"), (65536, 512, 128, "gelu"), (65536, 384, 128, "div8"),
          (32768, 512, 128, "gelu"), (32768, 128, 128, "res"), (32768, 384, 128, "none"), (524288, 128, 128, "none"),
          (524288, 128, 64, "mod64relu"), (524288, 64, 128, "none"), (32768, 256, 64, "none"), (32768, 64, 64, "none"), (8192, 512, 128, "gelu"),
          (8192, 256, 128, "none"), (524288, 128, 128, "res")]


def run(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M, N, K, mode in SHAPES:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    kw = dict(bias=b)
    if mode == "res":
        kw["aux0"] = torch.randn(M, N, device="cuda")
    elif mode == "gelu":
        kw["act"] = "gelu"
    elif mode == "div8":
        kw = dict(aux0=torch.randn(M // 8, N, device="cuda"), row_div=8)
    elif mode == "mod64relu":
        kw = dict(aux0=torch.randn(64, N, device="cuda"), row_mod=64, act="relu")
    outs, line = {}, f"M={M:>6} N={N:>3} K={K:>3} {mode:>9}:"
    fl = 2.0 * M * N * K
    for tile in (13, 20, 3):
        if tile == 13 and K < 128:
            continue
        o = torch.zeros(M, N, device="cuda")
        try:
            t = run(lambda: ops.conv_gemm(x, w, o, tile=tile, **kw))
        except Exception as ex:
            line += f" t{tile} n/a |"
            continue
        outs[tile] = o
        line += f" t{tile} {t:6.1f} {fl / t / 1e6:5.1f}TF |"
    ref = outs[3]
    line += " bitexact " + ",".join(f"{k}:{int(torch.equal(v, ref))}" for k, v in outs.items() if k != 3)
    print(line, flush=True)
    del outs, x

# LayerNorm prologue vs LayerNorm kernel + GEMM
for M, N in ((65536, 128), (65536, 512), (65536, 384)):
    x = torch.randn(M, 128, device="cuda") * 3 + 1
    g, be = torch.rand(128, device="cuda") + 0.5, torch.randn(128, device="cuda")
    w, b = torch.randn(N, 128, device="cuda") / 11, torch.randn(N, device="cuda")
    y, o1, o2 = torch.empty_like(x), torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    wf, bf = ops.fold_layernorm(g, be, w, b)
    t1 = run(lambda: (ops.layernorm(x, g, be, y, 1e-5), ops.conv_gemm(y, w, o1, bias=b, tile=13)))
    t1b = run(lambda: (ops.layernorm(x, g, be, y, 1e-5), ops.conv_gemm(y, w, o1, bias=b)))
    t2 = run(lambda: ops.conv_gemm(x, wf, o2, bias=bf, ln_eps=1e-5))
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x.double(), (128,), g.double(), be.double(), 1e-5), w.double(), b.double())
    print(f"LN+GEMM M={M} N={N}: LN kernel + t13 {t1:6.1f}us | LN kernel + auto {t1b:6.1f}us | fused {t2:6.1f}us | err unfused {(o1 - ref).abs().max().item():.2e} "
          f"fused {(o2 - ref).abs().max().item():.2e}", flush=True)

import sys, torch
sys.path.insert(0, '.')
import stitch_amd
ops = stitch_amd.ops
def run(fn, iters=40):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K, batch) in ((8192, 256, 160, 1), (8192, 128, 192, 1), (4096, 128, 128, 1), (8192, 256, 128, 1), (128, 4096, 128, 2), (8192, 64, 256, 1), (512, 256, 128, 1), (8192,128,224,1), (8192,256,1920,1)):
    x = torch.randn(batch * M, K, device="cuda"); w = torch.randn(batch * N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    ws = torch.empty(16 << 20, device="cuda")
    line = f"M={M} N={N} K={K} b{batch}:"
    outs = {}
    for tile in (3, 13, 14, 4):
        o = torch.zeros(batch * M, N, device="cuda")
        kw = dict(batch=batch, bsa=M * K, bsw=N * K, bsc=M * N, M=M) if batch > 1 else {}
        try:
            with ops.workspace_scope(ws):
                t = run(lambda: ops.conv_gemm(x, w[:N], o, bias=b, act="relu", tile=tile, split_k=1, **kw))
        except Exception as e:
            line += f" t{tile} n/a"; continue
        outs[tile] = o
        line += f" t{tile} {t:5.1f}us"
    line += " eq " + ",".join(str(int(torch.equal(v, outs[3]))) for v in outs.values())
    print(line, flush=True)

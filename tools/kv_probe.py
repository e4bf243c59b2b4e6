import sys, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
for (B, heads, Nq, Nk, D) in ((8, 8, 4096, 256, 16), (2, 4, 16384, 256, 32), (2, 8, 4096, 256, 32)):
    C = heads * D
    q = torch.randn(B, Nq, C, device="cuda"); k = torch.randn(B, Nk, C, device="cuda"); v = torch.randn(B, Nk, C, device="cuda")
    out = torch.empty(B, Nq, C, device="cuda")
    f = lambda: ops.attention_kvlds(q, (Nq * C, C), k, (Nk * C, C), v, (Nk * C, C), out, (Nq * C, C), B, heads, Nq, Nk, D, D ** -0.5)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} heads={heads} Nq={Nq} Nk={Nk} D={D}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")

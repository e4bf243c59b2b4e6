"""ST_MLP3_DIAG=1 python tools/mlp_split3_diag.py : per-phase cycle sums of the instrumented st_mlp128_split3 (M = 65536, hidden 512, with / without projection)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stitch_amd
ops = stitch_amd.ops
gen = torch.Generator().manual_seed(3)
for M, proj in ((65536, True), (65536, False), (32768, True)):
    hidden = 512
    x, x0 = torch.randn(M, 128, generator=gen).cuda(), torch.randn(M, 128, generator=gen).cuda()
    w1, b1 = (torch.randn(hidden, 128, generator=gen) / 128 ** 0.5).cuda(), (torch.randn(hidden, generator=gen) * 0.1).cuda()
    w2, b2 = (torch.randn(128, hidden, generator=gen) / hidden ** 0.5).cuda(), (torch.randn(128, generator=gen) * 0.1).cuda()
    wp, bp = (torch.randn(128, 128, generator=gen) / 128 ** 0.5).cuda(), (torch.randn(128, generator=gen) * 0.1).cuda()
    img = ops.mlp128_split3_pack(w1, b1, w2, proj=(wp, bp) if proj else None)
    o = torch.empty(M, 128, device="cuda")
    print(f"M {M} proj {proj}", file=sys.stderr, flush=True)
    for _ in range(3):
        ops.mlp128(x, o, w1, b1, w2, b2, ln_eps=1e-6, proj=(wp, bp, x0) if proj else None, image=img)
    torch.cuda.synchronize()

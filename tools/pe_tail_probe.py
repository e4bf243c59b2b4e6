"""st_pe_tail_split3 against the three launches it replaces (GPU box): python tools/pe_tail_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import stitch_amd
from tools.mlp_split3_probe import timed
ops = stitch_amd.ops
gen = torch.Generator().manual_seed(5)
R, P = 8192 * 64, 64
x = torch.randn(R, 64, generator=gen).cuda()
w1, tab = (torch.randn(128, 128, generator=gen) / 128 ** 0.5).cuda(), (torch.randn(P, 128, generator=gen) * 0.5).cuda()
w2, b2 = (torch.randn(128, 128, generator=gen) / 128 ** 0.5).cuda(), (torch.randn(128, generator=gen) * 0.1).cuda()
gam, bet = (torch.rand(128, generator=gen) + 0.5).cuda(), (torch.randn(128, generator=gen) * 0.1).cuda()
s4, tok, out = (torch.empty(R, 128, device="cuda") for _ in range(3))
img = ops.pe_tail_split3_pack(w1, w2)
w1h = w1[:, :64]

def three():
    ops.conv_gemm(x, w1h, s4, aux0=tab, row_mod=P, act="relu")
    ops.conv_gemm(s4, w2, tok, bias=b2)
    ops.layernorm(tok, gam, bet, tok, 1e-5)

te = timed(three, 20)
ts = timed(lambda: ops.pe_tail_split3(x, tab, img, b2, gam, bet, out), 20)
print(f"PatchEmbed tail, {R} rows: three fp32 launches {te:.1f} us, st_pe_tail_split3 {ts:.1f} us (x{te / ts:.2f}); 402 MB of rows = {402e6 / ts / 1e6:.2f} TB/s")

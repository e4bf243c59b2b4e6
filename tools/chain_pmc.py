import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
ops = stitch_amd.ops
M = 65536
att, x = torch.randn(M, 128, device="cuda"), torch.randn(M, 128, device="cuda")
ws = [(torch.randn(128, 128, device="cuda") / 11, torch.randn(128, device="cuda")) for _ in range(3)]
o2 = torch.empty(M, 128, device="cuda")
for _ in range(4):
    ops.linear_chain128(att, o2, [dict(w=ws[0][0], bias=ws[0][1], res=x), dict(w=ws[1][0], bias=ws[1][1], act="gelu", ln_eps=1e-5),
                                  dict(w=ws[2][0], bias=ws[2][1], res=1)])
torch.cuda.synchronize()

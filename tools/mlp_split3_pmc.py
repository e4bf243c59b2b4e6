"""5 launches each of st_mlp128 (fp32 MFMA) and st_mlp128_split3 on M = 65536, hidden 512, projection in front: the program
tools/mlp_split3_pmc.sh profiles (one rocprofv3 --pmc group per pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stitch_amd
ops = stitch_amd.ops
gen = torch.Generator().manual_seed(3)
M, hidden = 65536, 512
x, x0 = torch.randn(M, 128, generator=gen).cuda(), torch.randn(M, 128, generator=gen).cuda()
w1, b1 = (torch.randn(hidden, 128, generator=gen) / 128 ** 0.5).cuda(), (torch.randn(hidden, generator=gen) * 0.1).cuda()
w2, b2 = (torch.randn(128, hidden, generator=gen) / hidden ** 0.5).cuda(), (torch.randn(128, generator=gen) * 0.1).cuda()
wp, bp = (torch.randn(128, 128, generator=gen) / 128 ** 0.5).cuda(), (torch.randn(128, generator=gen) * 0.1).cuda()
img = ops.mlp128_split3_pack(w1, b1, w2, proj=(wp, bp))
o = torch.empty(M, 128, device="cuda")
for image in (None, img):
    for _ in range(5):
        ops.mlp128(x, o, w1, b1, w2, b2, ln_eps=1e-6, proj=(wp, bp, x0), image=image)
    torch.cuda.synchronize()

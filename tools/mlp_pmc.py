"""One st_mlp128 launch at the vertical layers' shape (M = 65536, hidden 512) for the SQ counter passes of tools/mlp_pmc.sh."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
ops = stitch_amd.ops
M = 65536
x = torch.randn(M, 128, device="cuda")
w1, b1 = torch.randn(512, 128, device="cuda") / 11, torch.randn(512, device="cuda") * 0.1
w2, b2 = torch.randn(128, 512, device="cuda") / 22, torch.randn(128, device="cuda") * 0.1
o = torch.empty(M, 128, device="cuda")
for _ in range(6):
    ops.mlp128(x, o, w1, b1, w2, b2, ln_eps=1e-6)
torch.cuda.synchronize()
